export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "dedup or config0" > gpurun_out/r03_dedupe_test.log 2>&1
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_round2.py -x -q > gpurun_out/r03_engine_tests.log 2>&1
for i in 1 2; do
DTP_NO_DEDUPE=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_dd0_b1_$i.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_dd1_b1_$i.log 2>&1
done
DTP_NO_DEDUPE=1 timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_dd0_b8.log 2>&1
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_dd1_b8.log 2>&1
