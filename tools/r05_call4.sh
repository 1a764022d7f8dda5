#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or cross_attention" > gpurun_out/r05_attn_tests.log 2>&1
tail -6 gpurun_out/r05_attn_tests.log
timeout 600 python tools/bench_attn.py > gpurun_out/r05_bench_attn_new.log 2>&1
cat gpurun_out/r05_bench_attn_new.log
bash tools/ab.sh tools/ab/libdtp_head.so all 2
