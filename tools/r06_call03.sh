#!/bin/bash
# round 6, call 3: fp64 GroupNorm reductions + GroupNorm-on-load proj_in: parity tests, then a same-box A/B against the round-5 library
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "groupnorm or gn or statistics or layernorm_fold or resident" 2>&1 | tail -15
timeout 600 python tools/scratch/gn_large_mean.py > gpurun_out/r06_gn_large_mean_after.log 2>&1; grep -v amdgpu gpurun_out/r06_gn_large_mean_after.log | tail -20
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -q -x -k "not batch16 and not trained_like and not fp8" 2>&1 | tail -8
rm -f gpurun_out/ab_summary.log
bash tools/ab.sh tools/ab/libdtp_r05.so b1 2
bash tools/ab.sh tools/ab/libdtp_r05.so 256 2
grep -h graph_nodes gpurun_out/ab_b1_ref_1.log gpurun_out/ab_b1_new_1.log | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['ms_per_step'], d['config']['graph_nodes'])"
