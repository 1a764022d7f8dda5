#!/bin/bash
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
for lib in tools/ab/libdtp_oldstats.so tools/ab/libdtp_oldapply.so; do
  echo "== lib=$lib: $(DTP_LIB=$lib timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k test_deduplicated_prefix_is_bit_identical 2>&1 | grep -E 'passed|failed|AssertionError: tensor' | tr '\n' ' ')"
done
