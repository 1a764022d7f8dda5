#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_streaming" 2>&1 | tail -5
timeout 600 python tools/diag_ws.py --cold --ws --noreduce 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/r04_diag_ws_nt2.log
timeout 900 python tools/diag_ws.py --big --noreduce 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee gpurun_out/r04_diag_ws_big.log
