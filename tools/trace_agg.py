#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid size): count, avg/min duration.  Usage: trace_agg.py trace.csv out.csv"""
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 1e30])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", r.get("Name", "?"))
        for pre in ("void (anonymous namespace)::", "(anonymous namespace)::"):
            name = name.replace(pre, "")
        key = (name[:70], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Z", "?"), r.get("LDS_Block_Size", "?"))
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg[key]
        a[0] += 1; a[1] += d; a[2] = min(a[2], d)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_x", "grid_z", "lds", "calls", "total_us", "avg_us", "min_us"])
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow(list(k) + [a[0], round(a[1], 1), round(a[1] / a[0], 2), round(a[2], 2)])
