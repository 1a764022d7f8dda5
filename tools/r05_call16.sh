#!/bin/bash
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "conv" > gpurun_out/r05_cw_ops.log 2>&1; tail -3 gpurun_out/r05_cw_ops.log
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x > gpurun_out/r05_cw_engine.log 2>&1; tail -3 gpurun_out/r05_cw_engine.log
bash tools/ab.sh tools/ab/libdtp_gh.so all 3
