#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv by kernel name: sum and per-dispatch mean of each counter.
Usage: pmc_agg.py counter_collection.csv out.csv"""
import csv, re, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", "?")
        for pre in ("void (anonymous namespace)::", "(anonymous namespace)::"):
            name = name.replace(pre, "")
        m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)  # kernels whose signature names _Float16 stay mangled: keep the bare kernel name
        if m:
            name = name[m.end():m.end() + int(m.group(1))] + " [" + name[m.end() + int(m.group(1)):][:40] + "]"
        a = agg[name[:80]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "sum", "mean_per_dispatch"])
    for k, cs in sorted(agg.items(), key=lambda kv: -max(v[1] for v in kv[1].values())):
        for c, a in cs.items():
            w.writerow([k, c, a[0], a[1], a[1] / a[0]])
