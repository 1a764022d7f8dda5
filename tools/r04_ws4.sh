#!/bin/bash
mkdir -p gpurun_out
for v in "" nomfma noldsread nowload nodma; do
  echo "== variant '${v:-full}' with DTP_WS_DEBUG=2 (no combine / store): main loop only, cold weights"
  if [ -n "$v" ]; then export DTP_LIB=tools/ab/libdtp_ws_$v.so; else unset DTP_LIB; fi
  DTP_WS_DEBUG=2 timeout 600 python tools/diag_ws.py --cold --ws --noreduce 2>&1 | grep -v amdgpu.ids | cut -c1-250
done > gpurun_out/r04_diag_ws_variants.log 2>&1
cat gpurun_out/r04_diag_ws_variants.log
