#!/usr/bin/env python3
"""Compare two --dump-launches CSVs (kind,us,tflops,algo_GBps,label): total us per shape label (tile / split stripped)."""
import csv, sys, collections, re
def load(p):
    d = collections.defaultdict(lambda: [0, 0.0, ""])
    for r in csv.reader(open(p)):
        if len(r) < 5 or r[0] == "kind": continue
        lab = r[4]
        key = re.sub(r" tile=\d+ splits=\d+", "", lab)
        m = re.search(r"tile=(\d+) splits=(\d+)", lab)
        e = d[key]; e[0] += 1; e[1] += float(r[1]); e[2] = m.group(0) if m else ""
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in set(a) | set(b):
    ea, eb = a.get(k, [0, 0.0, ""]), b.get(k, [0, 0.0, ""])
    rows.append((ea[1], eb[1], ea[0], k, ea[2], eb[2]))
rows.sort(reverse=True)
ta = sum(r[0] for r in rows); tb = sum(r[1] for r in rows)
print(f"total A {ta/1e3:.2f} ms  B {tb/1e3:.2f} ms")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f"{r[0]:9.1f} {r[1]:9.1f} {100*(r[1]-r[0])/max(r[0],1e-9):+6.1f}% n={r[2]:4d} {r[3]}  [{r[4]}] [{r[5]}]")
