#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or cross_attention" > gpurun_out/r05_attn_tests.log 2>&1
tail -4 gpurun_out/r05_attn_tests.log
bash tools/r05_dumpcmp.sh 2>&1 | grep -i "total\|attn\|M=4096 N=320 K=128\|M=4096 N=128"
bash tools/ab.sh tools/ab/libdtp_head.so all 2
