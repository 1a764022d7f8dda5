#!/bin/bash
# lnlin start skew: launch times per skew value, then the whole-stamp A/B for the best one
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
for k in 0 1 2 3 4 6 8 12; do echo "skew $k: $(DTP_LNLIN_SKEW=$k timeout 300 python tools/bench_lnlin.py --child 2>&1 | grep -v amdgpu.ids)"; done | tee gpurun_out/r05_lnlin_skew.log
