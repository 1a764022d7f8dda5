#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "weight_streaming" 2>&1 | tail -5
for dbg in 0 1 2 3; do
  echo "== DTP_WS_DEBUG=$dbg (1: no main loop, 2: no combine/store), kernel only (NOREDUCE), cold weights"
  DTP_WS_DEBUG=$dbg timeout 600 python tools/diag_ws.py --cold --ws --noreduce 2>&1 | grep -v amdgpu.ids | cut -c1-330
done > gpurun_out/r04_diag_ws_dbg2.log 2>&1
cat gpurun_out/r04_diag_ws_dbg2.log
