#!/bin/bash
# three-images-per-workgroup halo conv: op parity, then same-box A/B of the tuned stamp (separate tune caches)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "halo" > gpurun_out/r03_ops11.log 2>&1
export DTP_TUNE_REPORT=1
DTP_NO_HALO3=1 DTP_TUNE_CACHE=/tmp/tcA.txt timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_h3a_tune.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_h3b_tune.log 2>&1
unset DTP_TUNE_REPORT
for i in 1 2 3; do
DTP_NO_HALO3=1 DTP_TUNE_CACHE=/tmp/tcA.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_h3a_b1_$i.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_h3b_b1_$i.log 2>&1
done
DTP_NO_HALO3=1 DTP_TUNE_CACHE=/tmp/tcA.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_h3a_256.log 2>&1
DTP_TUNE_CACHE=/tmp/tcB.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_h3b_256.log 2>&1
grep -c " 48 \| 49 " /tmp/tcB.txt > gpurun_out/r03_h3_count.txt; grep " 48 \| 49 " /tmp/tcB.txt >> gpurun_out/r03_h3_count.txt
DTP_TUNE_CACHE=/tmp/tcB.txt DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0 or 256_10steps" > gpurun_out/r03_parity11.log 2>&1
