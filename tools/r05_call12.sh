#!/bin/bash
# round-4 HEAD build (6f37f17, rebuilt from git) against the round-5 final build, one box, separate tune tables (both start from the shipped one)
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc_new.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
AB_ENV="DTP_TUNE_CACHE=/tmp/ab_tc_new.txt" bash tools/ab.sh tools/ab/libdtp_r04.so all 3
wc -l /tmp/ab_tc.txt /tmp/ab_tc_new.txt
