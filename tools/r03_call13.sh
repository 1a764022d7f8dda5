#!/bin/bash
# gemm_kernel DMA as buffer loads: op parity, stamp A/B against the previous build (same tune-cache key: one cache per arm)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r03_ops13.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so timeout 300 python tools/diag_lw.py > gpurun_out/r03_diag_lw_head.log 2>&1
timeout 300 python tools/diag_lw.py > gpurun_out/r03_diag_lw_new.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tcA.txt timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sa_tune.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sb_tune.log 2>&1
for i in 1 2 3; do
DTP_LIB=tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tcA.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sa_b1_$i.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sb_b1_$i.log 2>&1
done
DTP_LIB=tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tcA.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sa_256.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sb_256.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tcA.txt timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sa_b8.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_sb_b8.log 2>&1
cp /tmp/tcB.txt gpurun_out/r03_tcB.txt
DTP_TUNE_CACHE=/tmp/tcB.txt DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0 or 256_10steps" > gpurun_out/r03_parity13.log 2>&1
