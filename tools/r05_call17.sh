#!/bin/bash
mkdir -p gpurun_out
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
for lib in "" tools/ab/libdtp_gh.so tools/ab/libdtp_gnstats.so tools/ab/libdtp_head.so; do
  for rep in 1 2; do
    echo "== lib=${lib:-working} rep $rep: $(DTP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k test_deduplicated_prefix_is_bit_identical 2>&1 | grep -E 'passed|failed|AssertionError: tensor' | tr '\n' ' ')"
  done
done
