export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "groupnorm or attention" > gpurun_out/r03_ops2.log 2>&1
for p in 0 1 3; do DTP_ATTN_PRIO=$p timeout 300 python tools/diag_attn.py >> gpurun_out/r03_attn_prio.log 2>&1; done
timeout 600 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r03_rg_b1.log 2>gpurun_out/r03_rg_b1.err
DTP_NO_FUSE_REDUCE_GN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_rg0_b1.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_rg1_b1.log 2>&1
DTP_NO_FUSE_REDUCE_GN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_rg0_b1_2.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > gpurun_out/r03_rg1_b1_2.log 2>&1
