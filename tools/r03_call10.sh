export DTP_TUNE_SEED=/tmp/none.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "splitk or conv3x3 or halo or two_activation" > gpurun_out/r03_ops8.log 2>&1
DTP_NO_XCD_SPLIT=1 DTP_TUNE_CACHE=/tmp/tc_x0.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
DTP_TUNE_CACHE=/tmp/tc_x1.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
for i in 1 2; do
DTP_NO_XCD_SPLIT=1 DTP_TUNE_CACHE=/tmp/tc_x0.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xs0_b1_$i.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_x1.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xs1_b1_$i.log 2>&1
done
DTP_NO_XCD_SPLIT=1 DTP_TUNE_CACHE=/tmp/tc_x0.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xs0_256.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_x1.txt timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xs1_256.log 2>&1
DTP_NO_XCD_SPLIT=1 DTP_TUNE_CACHE=/tmp/tc_x0.txt timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xs0_b8.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_x1.txt timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xs1_b8.log 2>&1
DTP_TUNE_CACHE=/tmp/tc_x1.txt timeout 600 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r03_launches_b1_xs.csv > gpurun_out/r03_xs_b1_prof.log 2>&1
cp /tmp/tc_x1.txt gpurun_out/r03_tc_xs1.txt; cp /tmp/tc_x0.txt gpurun_out/r03_tc_xs0.txt
