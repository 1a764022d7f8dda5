#!/bin/bash
# round 6: test_gpu_round2 in full with the runtime configuration of the package (the suite run of r06_call29.sh stopped at its asynchrony test), host cost per stamp with and without captured packets
mkdir -p gpurun_out
( time DTP_RUNTIME_ENV=1 timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -s -k "enqueue" ) 2>&1 | grep "host enqueue\|passed\|failed" | tee gpurun_out/r06_host_enqueue.txt
DTP_RUNTIME_ENV=0 timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -s -k "enqueue" 2>&1 | grep "host enqueue\|passed\|failed" | sed 's/^/captured packets (runtime default): /' | tee -a gpurun_out/r06_host_enqueue.txt
( time DTP_RUNTIME_ENV=1 timeout 1200 python -m pytest tests/test_gpu_round2.py -q -m gpu -x ) 2>&1 | tail -5
