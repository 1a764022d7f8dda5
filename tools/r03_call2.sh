export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r03_ops_tests.log 2>&1
( time timeout 1500 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r03_launches_b1_lw.csv > gpurun_out/r03_lw_b1.log 2>gpurun_out/r03_lw_b1.err ) 2> gpurun_out/r03_lw_b1.time
cp /tmp/tc.txt gpurun_out/r03_tc_lw.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_lw_b1_again.log 2>&1
