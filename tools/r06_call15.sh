#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "groupnorm or gn or statistics" 2>&1 | tail -3
rm -f gpurun_out/ab_summary.log
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
bash tools/ab.sh tools/ab/libdtp_r05.so 256 4
bash tools/ab.sh tools/ab/libdtp_r05.so b1 2
