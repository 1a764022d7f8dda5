#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_streaming" 2>&1 | tail -5
timeout 600 python tools/diag_ws.py --cold --tail --noreduce 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee gpurun_out/r04_diag_ws_tail.log
