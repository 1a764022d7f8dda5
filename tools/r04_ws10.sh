#!/bin/bash
mkdir -p gpurun_out
cp tools/ab/tc_ws.txt /tmp/ab_tc.txt
bash tools/ab.sh tools/ab/libdtp_head.so all 2
cp /tmp/ab_tc.txt gpurun_out/r04_tc_ws2.txt
grep ",ws2 " /tmp/ab_tc.txt | awk '{print $2}' | sort | uniq -c
