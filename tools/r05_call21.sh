#!/bin/bash
# long-shortcut / residual convs of level 0 on convws tile 54 (statistics from the epilogue) instead of the halo kernel the tuner measured
# 1-2 us faster: same library, two tune tables
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
cp tools/ab/seed_mod.txt /tmp/ab_tc_mod.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
AB_ENV="DTP_TUNE_CACHE=/tmp/ab_tc_mod.txt" bash tools/ab.sh diffusiontexturepainting_amd/libdtp.so all 3
grep -h -o '"graph_nodes": [0-9]*' gpurun_out/ab_b1_ref_1.log gpurun_out/ab_b1_new_1.log gpurun_out/ab_b8_ref_1.log gpurun_out/ab_b8_new_1.log
