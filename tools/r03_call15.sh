#!/bin/bash
# lnlin_kernel: op parity, per-shape timing, stamp A/B against the previous build
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "lnlin or layernorm_fold" > gpurun_out/r03_ops15.log 2>&1
timeout 400 python tools/diag_lnlin.py > gpurun_out/r03_diag_lnlin.log 2>&1
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
DTP_TUNE_CACHE=/tmp/tcB.txt DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0 or 256_10steps" > gpurun_out/r03_parity15.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 900 python -m pytest tests/test_gpu_engine.py -x -q > gpurun_out/r03_engine15.log 2>&1
