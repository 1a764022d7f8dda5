#!/bin/bash
# round 4, first GPU call: parity of convws_kernel, then its timing against the tuned tiles (hot and cold weights)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_streaming" > gpurun_out/r04_ws_ops.log 2>&1
tail -5 gpurun_out/r04_ws_ops.log
timeout 900 python tools/diag_ws.py > gpurun_out/r04_diag_ws_hot.log 2>&1
timeout 900 python tools/diag_ws.py --cold > gpurun_out/r04_diag_ws_cold.log 2>&1
tail -40 gpurun_out/r04_diag_ws_cold.log
