#!/bin/bash
# gn_apply_kernel: the first two items' loads issued before the statistics prologue.  Op parity + stamp A/B (shipped tune table)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm or group_norm" > gpurun_out/r03_ops24.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so timeout 300 python tools/diag_gn.py > gpurun_out/r03_diag_gn24_head.log 2>&1
timeout 300 python tools/diag_gn.py > gpurun_out/r03_diag_gn24.log 2>&1
export DTP_TUNE_CACHE=/tmp/tc.txt
for i in 1 2 3; do
DTP_LIB=tools/ab/libdtp_head.so timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ga_b1_$i.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gb_b1_$i.log 2>&1
done
DTP_LIB=tools/ab/libdtp_head.so timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ga_b8.log 2>&1
timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gb_b8.log 2>&1
