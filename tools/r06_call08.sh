#!/bin/bash
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py -q -x -k "not batch16 and not trained_like and not fp8" 2>&1 | tail -5
rm -f gpurun_out/ab_summary.log
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
bash tools/ab.sh tools/ab/libdtp_r05.so all 2
