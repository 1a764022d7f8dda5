#!/bin/bash
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
echo "== default"; timeout 600 python tools/scratch/repro_z.py --u3first 2>&1 | grep -v amdgpu.ids | tail -6
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py tests/test_gpu_round2.py -q -x > gpurun_out/r05_rgn_tests2.log 2>&1
tail -4 gpurun_out/r05_rgn_tests2.log
