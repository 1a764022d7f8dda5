"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel name, grid size) launches, min / median duration in us.
Usage: python tools/ktrace_summary.py <dir-or-csv> [name-substring ...]"""
import csv, glob, os, sys, statistics, re

def main():
    path = sys.argv[1]
    pats = sys.argv[2:]
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name)[:70]
            if pats and not any(p in name for p in pats):
                continue
            key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
            rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for (name, grid), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print(f"{len(d):5d} x  min {min(d):7.1f}  med {statistics.median(d):7.1f} us  grid {grid:>8}  {name}")

if __name__ == "__main__":
    main()
