#!/bin/bash
# xattn_kernel with a 4-deep phase-1 ring: op parity, stamp A/B (512^2 and 256^2) against the previous build; shipped tune table
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "cross_attention" > gpurun_out/r03_ops20.log 2>&1
export DTP_TUNE_CACHE=/tmp/tc.txt
for i in 1 2 3; do
DTP_LIB=tools/ab/libdtp_head.so timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xa_b1_$i.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xb_b1_$i.log 2>&1
done
for i in 1 2; do
DTP_LIB=tools/ab/libdtp_head.so timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xa_256_$i.log 2>&1
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xb_256_$i.log 2>&1
done
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0 or 256_10steps" > gpurun_out/r03_parity20.log 2>&1
