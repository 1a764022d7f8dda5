# regenerate the shipped tune table: every shape the bench (B=1 with extras, B=8) and the GPU test-suite build, tuned on this box
export DTP_TUNE_CACHE=/tmp/tc.txt
export DTP_TUNE_SEED=/tmp/none.txt
rm -f /tmp/tc.txt
timeout 1500 python bench.py > gpurun_out/r03_seed_b1.log 2>gpurun_out/r03_seed_b1.err
timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r03_seed_b8.log 2>&1
( time timeout 2400 python -m pytest tests -q -m gpu --durations=25 -x ) > gpurun_out/r03_seed_tests.log 2>&1
cp /tmp/tc.txt gpurun_out/r03_tune_seed.txt
wc -l /tmp/tc.txt
