#!/bin/bash
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
echo "== default"; timeout 600 python tools/scratch/repro_z.py --u3first 2>&1 | grep -v amdgpu.ids | tail -9
echo "== concat-reduce off"; DTP_NO_REDUCE_IN_CONCAT_GN=1 timeout 600 python tools/scratch/repro_z.py --u3first 2>&1 | grep -v amdgpu.ids | tail -9
echo "== lnlin ablation"; timeout 900 python tools/bench_lnlin.py 2>&1 | tee gpurun_out/r05_lnlin_ablation_raw.log
