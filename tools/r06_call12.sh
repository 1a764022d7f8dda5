#!/bin/bash
# round 6: six-slot rings in xchain_kernel / ffchain_kernel against the three-slot build (tools/ab/libdtp_r06a.so = HEAD before the change)
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "register_chained" 2>&1 | tail -4
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "deduplicated or config1_512_20steps_properties or batch8" 2>&1 | tail -3
rm -f gpurun_out/ab_summary.log
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
bash tools/ab.sh tools/ab/libdtp_r06a.so b1 3
bash tools/ab.sh tools/ab/libdtp_r06a.so b8 2
AB_ENV="DTP_FFCHAIN=1" bash tools/ab.sh tools/ab/libdtp_r06a.so b1 1
cp gpurun_out/ab_summary.log gpurun_out/r06_ab_ring6.txt
