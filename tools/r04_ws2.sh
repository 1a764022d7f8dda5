#!/bin/bash
mkdir -p gpurun_out
for dbg in 0 1 2 3 4 8; do
  echo "== DTP_WS_DEBUG=$dbg (1: no main loop, 2: no combine/store, 4: nt off, 8: nt on), kernel only (NOREDUCE), cold weights"
  DTP_WS_DEBUG=$dbg timeout 600 python tools/diag_ws.py --cold --ws --noreduce 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r04_diag_ws_dbg.log 2>&1
cat gpurun_out/r04_diag_ws_dbg.log
