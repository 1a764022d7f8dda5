#!/bin/bash
# gn_grid_kernel: what its barrier costs (ablation bits, launch times in us), then parity of the two barrier forms
mkdir -p gpurun_out
{
DTP_NO_GN_GRID=1 timeout 300 python tools/bench_gn_grid.py
for m in 0 16 15 31 21 24 8 2; do DTP_GN_GRID_MODE=$m timeout 300 python tools/bench_gn_grid.py; done
} 2>&1 | grep -v Warning | tee gpurun_out/r06_gn_grid_ablation.txt
for m in 0 16; do
  echo "== parity, mode $m"
  DTP_GN_GRID_MODE=$m timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "groupnorm_grid or groupnorm_result" 2>&1 | grep -v "^$" | tail -25
done 2>&1 | tee gpurun_out/r06_gn_grid_parity.txt
