#!/bin/bash
mkdir -p gpurun_out
for v in "" nomfma noldsread nowload nodma; do
  echo "== variant '${v:-full}'"
  if [ -n "$v" ]; then export DTP_LIB=tools/ab/libdtp_ws_$v.so; else unset DTP_LIB; fi
  timeout 600 python tools/diag_ws.py --big --ws --noreduce 2>&1 | grep -v amdgpu.ids | grep "2wg" | cut -c1-120
done > gpurun_out/r04_diag_ws_variants_big.log 2>&1
cat gpurun_out/r04_diag_ws_variants_big.log
