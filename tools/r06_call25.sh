#!/bin/bash
# the single-launch GroupNorm (gn_fused_kernel: one block per (sample, group slab), second pass from L2) on maps above 256 pixels: launch times per slab-size cap, then stamp A/B
mkdir -p gpurun_out
for kb in 0 100 260 400 1000; do DTP_GN_FUSED_KB=$kb timeout 300 python tools/bench_gn_grid.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_gn_fused_large_maps.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['config']['graph_nodes'])"; }
for i in 1 2; do for kb in 0 260 400; do
  DTP_GN_FUSED_KB=$kb timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | line "b1 kb=$kb"
  DTP_GN_FUSED_KB=$kb timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile --batch 8 --steps 3 --warmup 1 2>/dev/null | line "b8 kb=$kb"
done; done 2>&1 | tee -a gpurun_out/r06_gn_fused_large_maps.txt
