export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "cross_attention or batched or halo" > gpurun_out/r03_ops7.log 2>&1
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q > gpurun_out/r03_engine7.log 2>&1
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config0 or 256_20steps" > gpurun_out/r03_parity7.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
for i in 1 2; do
DTP_NO_GN_CONV=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc0_b1_$i.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc1_b1_$i.log 2>&1
done
DTP_NO_GN_CONV=1 timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc0_256.log 2>&1
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc1_256.log 2>&1
DTP_NO_XATTN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_xb0_b1.log 2>&1
timeout 600 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_gc1_b8.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r03_launches_b1_gc.csv > gpurun_out/r03_gc_b1_prof.log 2>&1
