#!/bin/bash
mkdir -p gpurun_out
cp tools/ab/tc_ws.txt /tmp/tc.txt
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 900 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r04_launches_b1.csv > gpurun_out/r04_dump_b1.log 2>&1
tail -1 gpurun_out/r04_dump_b1.log | cut -c1-300
