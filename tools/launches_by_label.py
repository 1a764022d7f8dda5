"""bench.py --dump-launches CSV (one line per profiled launch: kind, us, tflops, algo_GBps, label) -> one line per label:
kind, label, launches, total_ms, avg_us, tflops_last, algo_GBps_last.   python tools/launches_by_label.py in.csv out.csv"""
import csv, sys
from collections import OrderedDict
rows = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["kind"], r["label"])
    e = rows.setdefault(k, [0, 0.0, r["tflops"], r["algo_GBps"]])
    e[0] += 1; e[1] += float(r["us"]); e[2] = r["tflops"]; e[3] = r["algo_GBps"]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kind", "label", "launches", "total_ms", "avg_us", "tflops_last", "algo_GBps_last"])
    for (kind, label), (n, us, tf, gb) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        w.writerow([kind, label, n, "%.3f" % (us / 1e3), "%.2f" % (us / n), tf, gb])
