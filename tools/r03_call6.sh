timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q > gpurun_out/r03_ops4.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_new.txt timeout 900 python -m pytest tests/test_gpu_engine.py -x -q > gpurun_out/r03_engine4.log 2>&1
for i in 1 2; do
DTP_LIB=$PWD/tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tc_head.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ab_head_b1_$i.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_new.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ab_new_b1_$i.log 2>&1
done
DTP_LIB=$PWD/tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tc_head.txt timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ab_head_b8.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_new.txt timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ab_new_b8.log 2>&1
DTP_LIB=$PWD/tools/ab/libdtp_head.so DTP_TUNE_CACHE=/tmp/tc_head.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ab_head_256.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_new.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_ab_new_256.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_new.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r03_launches_b1_new.csv > gpurun_out/r03_new_b1_prof.log 2>&1
