#!/bin/bash
# round 6: every latency-path __shfl_xor replaced by DPP / permlane moves -- op tests, engine tests, A/B against the previous build
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -3
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py -q -x -k "not batch16 and not trained_like and not fp8 and not config0" 2>&1 | tail -3
rm -f gpurun_out/ab_summary.log
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
bash tools/ab.sh tools/ab/libdtp_r06b.so all 3
cp gpurun_out/ab_summary.log gpurun_out/r06_ab_dpp_shuffles.txt
