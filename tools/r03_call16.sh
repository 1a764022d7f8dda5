#!/bin/bash
# gemm_wide_kernel DMA as buffer loads: op parity, stamp A/B (batch 1 and 8) against the previous build; same tune-cache key
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r03_ops16.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tcA.txt timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sa_tune.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sb_tune.log 2>&1
for i in 1 2; do
DTP_LIB=tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tcA.txt timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sa_b8_$i.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sb_b8_$i.log 2>&1
done
for i in 1 2; do
DTP_LIB=tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tcA.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sa_b1_$i.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sb_b1_$i.log 2>&1
done
cp /tmp/tcB.txt gpurun_out/r03_tcB.txt
