#!/bin/bash
# GroupNorm apply on the halo conv's staged patch (GF_GNAPPLY), re-measured with the slim halo kernel: parity + stamp A/B
mkdir -p gpurun_out
cp gpurun_out/r03_tcB.txt /tmp/tc.txt 2>/dev/null
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
DTP_GN_CONV=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc0_b1_$i.log 2>&1
DTP_GN_CONV=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc1_b1_$i.log 2>&1
done
DTP_GN_CONV=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r03_launches_b1_gc.csv > gpurun_out/r03_gc1_b1_prof.log 2>&1
DTP_GN_CONV=1 timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc0_b8.log 2>&1
DTP_GN_CONV=1 timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc1_b8.log 2>&1
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "conv_input_patch" > gpurun_out/r03_parity17.log 2>&1
