#!/bin/bash
# per-launch tables of one profiled stamp for the reference build ($1) and the working build on the same box, then an interleaved A/B
REF=${1:-tools/ab/libdtp_r05.so}; RUNS=${2:-2}; WHAT=${3:-b1}
mkdir -p gpurun_out
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
DTP_LIB=$REF timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r06_dump_ref_b1.csv > gpurun_out/r06_dump_ref_b1.log 2>&1
env $AB_ENV timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r06_dump_new_b1.csv > gpurun_out/r06_dump_new_b1.log 2>&1
python tools/dump_cmp.py gpurun_out/r06_dump_ref_b1.csv gpurun_out/r06_dump_new_b1.csv 400 > gpurun_out/r06_dumpcmp_b1.txt
python - <<'P'
# only the labels that changed by more than 0.05 ms in total or exist on one side only
import re
for ln in open("gpurun_out/r06_dumpcmp_b1.txt"):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)", ln)
    if not m: print(ln.rstrip()); continue
    a, b = float(m.group(1)), float(m.group(2))
    if abs(a - b) > 50.0 or a == 0 or b == 0: print(ln.rstrip())
P
rm -f gpurun_out/ab_summary.log
bash tools/ab.sh $REF $WHAT $RUNS
