#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_streaming" 2>&1 | tail -5
cp tools/ab/tc_ws2.txt /tmp/ab_tc.txt 2>/dev/null
bash tools/ab.sh tools/ab/libdtp_head.so b1 2
cp /tmp/ab_tc.txt gpurun_out/r04_tc_ws3.txt
grep "ws2 " /tmp/ab_tc.txt | grep ",49,\|,53,\|,57," | head
