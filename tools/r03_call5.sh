export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm or attention" > gpurun_out/r03_ops3.log 2>&1
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0 or 256_20steps or dedup" > gpurun_out/r03_fold_parity.log 2>&1
for i in 1 2; do
DTP_NO_FOLD_GN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_fg0_b1_$i.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_fg1_b1_$i.log 2>&1
done
DTP_NO_FOLD_GN=1 timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_fg0_b8.log 2>&1
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_fg1_b8.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r03_launches_b1_fold.csv > gpurun_out/r03_fold_b1_prof.log 2>&1
