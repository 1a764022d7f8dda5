#!/bin/bash
# experimental weight prefetch on a forked graph branch ($DTP_PREFETCH = lead in contractions): does the replayed graph run it
# concurrently, and does the stamp get faster?  Same build, same box, shipped tune table.
mkdir -p gpurun_out
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_pf0_warm.log 2>&1
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_pf0_b1_$i.log 2>&1
DTP_PREFETCH=2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_pf2_b1_$i.log 2>&1
DTP_PREFETCH=5 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_pf5_b1_$i.log 2>&1
done
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_pf0_256.log 2>&1
DTP_PREFETCH=2 timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_pf2_256.log 2>&1
DTP_PREFETCH=2 DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0" > gpurun_out/r03_parity23.log 2>&1
