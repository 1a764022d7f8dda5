#!/usr/bin/env python3
"""Idle time between consecutive kernels of the graph-replayed stamps in a rocprofv3 kernel_trace.csv:
gap_analysis.py trace.csv  ->  busy / idle totals of the last replayed stamp and the gap histogram."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name", "?")))
rows.sort()
# the timed stamps are the last ones: take the final N kernels that belong to one stamp (step_kernel count = evals)
n_nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 6235
last = rows[-n_nodes:]
busy = sum(e - s for s, e, _ in last)
span = last[-1][1] - last[0][0]
gaps = [max(0, last[i + 1][0] - last[i][1]) for i in range(len(last) - 1)]
over = sum(1 for i in range(len(last) - 1) if last[i + 1][0] < last[i][1])
print(f"kernels {len(last)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {sum(gaps)/1e6:.2f} ms  overlapping pairs {over}")
h = collections.Counter()
for g in gaps:
    h[min(int(g / 500), 10)] += 1
for k in sorted(h):
    print(f"  gap {k*0.5:4.1f}-{k*0.5+0.5:4.1f} us: {h[k]}")
per = collections.defaultdict(lambda: [0, 0])
for (s, e, n), g in zip(last[1:], gaps):
    key = n.split("(")[0][-40:]
    per[key][0] += 1; per[key][1] += g
print("largest total gap BEFORE kernel:")
for k, (c, g) in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {g/1e3:9.1f} us over {c:5d} launches ({g/c/1e3:.2f} us each)  {k}")
