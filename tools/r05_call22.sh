#!/bin/bash
# after the hand-switched tune entries (level-0 long-shortcut / residual convs at batch 1 on convws tile 54: statistics from the epilogue):
# the batch-1 bench line, its rocprofv3 kernel statistics and launch table again, and the full-size tests that build those programs
mkdir -p gpurun_out
export DTP_ROUND=r05
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/tc.txt
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "config0 or properties or dedup or batch8 or batch16" > gpurun_out/r05_seed54_tests.log 2>&1; tail -2 gpurun_out/r05_seed54_tests.log
timeout 1500 python bench.py --dump-launches gpurun_out/r05_launches_b1.csv > gpurun_out/r05_b1.log 2>gpurun_out/r05_b1.err
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras > gpurun_out/r05_256.log 2>gpurun_out/r05_256.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r05 -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-extras > /root/repo/gpurun_out/r05_prof.log 2>&1
find /tmp/prof -name "*kernel_stats*" -exec cp {} /root/repo/gpurun_out/r05_kernel_stats.csv \;
cd /root/repo
wc -l /tmp/tc.txt
for f in b1 256; do grep "^{" gpurun_out/r05_$f.log | tail -1 | grep -o "\"ms_per_step\": [0-9.]*"; done
