#!/bin/bash
# round 6: the GPU suite on the final build with the shipped tune table (timed as the driver runs it), then what the driver does at round end
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -x -s -k config4 2>&1 | grep "fp8\|passed\|failed" | tee gpurun_out/r06_config4_fp8.txt
( time timeout 1500 python -m pytest tests -q -m gpu --durations=15 -x ) > gpurun_out/r06_gpu_suite.log 2>&1
tail -4 gpurun_out/r06_gpu_suite.log
bash tools/driver_style_check.sh 2>&1 | tail -12
