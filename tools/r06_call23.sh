#!/bin/bash
# round 6, final build: the profile refresh (without the two stand-alone CPU-oracle legs) + same-box A/B against the round-5 library + the GPU suite
mkdir -p gpurun_out
DTP_REFRESH_SKIP_ORACLE=1 DTP_REFRESH_AB=tools/ab/libdtp_r05.so bash tools/refresh_profiles_r06.sh
cat gpurun_out/r06_ab_r05_vs_r06_final.txt
