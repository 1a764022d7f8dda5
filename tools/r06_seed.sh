#!/bin/bash
# regenerate the shipped tune table: the shipped table first (entries of shapes whose candidate set did not change stay), then every
# shape the bench (B = 1 with extras -- now incl. batch 16 --, B = 8, 256^2) and the GPU test-suite build, tuned on this box
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/tc.txt
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 2400 python bench.py --no-cpu-baseline > gpurun_out/r06_seed_b1.log 2>gpurun_out/r06_seed_b1.err
timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r06_seed_b8.log 2>&1
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras > gpurun_out/r06_seed_256.log 2>&1
# first pass: every shape of the test-suite gets tuned (a comparison between two processes that tune the same shape concurrently may
# fail in this pass: they can pick different tiles); second pass: the suite as the driver runs it, on the complete table
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06_seed_tests_pass1.log 2>&1
( time timeout 2400 python -m pytest tests -q -m gpu --durations=25 -x ) > gpurun_out/r06_seed_tests.log 2>&1
cp /tmp/tc.txt gpurun_out/r06_tune_seed.txt
wc -l /tmp/tc.txt
tail -5 gpurun_out/r06_seed_tests.log
tail -1 gpurun_out/r06_seed_b1.log | cut -c1-300
