#!/bin/bash
# slab sums: the last (up to three) slabs as one predicated group of loads; reduce + GroupNorm: all items' slab sums before the first
# store.  Op parity (bit-identical sums expected), stamp A/B against the previous build, shipped tune table
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm or reduce or split or halo or conv3x3" > gpurun_out/r03_ops21.log 2>&1
timeout 300 python tools/diag_rgn.py > gpurun_out/r03_diag_rgn21.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so timeout 300 python tools/diag_rgn.py > gpurun_out/r03_diag_rgn21_head.log 2>&1
export DTP_TUNE_CACHE=/tmp/tc.txt
for i in 1 2 3; do
DTP_LIB=tools/ab/libdtp_head.so timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ra_b1_$i.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_rb_b1_$i.log 2>&1
done
for i in 1 2; do
DTP_LIB=tools/ab/libdtp_head.so timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ra_256_$i.log 2>&1
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_rb_256_$i.log 2>&1
done
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0 or 256_10steps or dedup" > gpurun_out/r03_parity21.log 2>&1
