#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_streaming" 2>&1 | tail -5
bash tools/ab.sh tools/ab/libdtp_head.so b1 2
bash tools/ab.sh tools/ab/libdtp_head.so 256 2
cp /tmp/ab_tc.txt gpurun_out/r04_tc_ws.txt
grep -c . /tmp/ab_tc.txt
grep ",ws " /tmp/ab_tc.txt
tail -3 gpurun_out/ab_b1_new_1.log | cut -c1-600
