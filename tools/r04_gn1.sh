#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_streaming" 2>&1 | tail -5
cp tools/ab/tc_ws.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -5
for i in 1 2; do
DTP_NO_GN_EPILOGUE=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('off', d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('on ', d['ms_per_step'])"
done
