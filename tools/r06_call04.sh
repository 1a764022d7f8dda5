#!/bin/bash
mkdir -p gpurun_out
echo "== r05 library"; DTP_LIB=tools/ab/libdtp_r05.so timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "deduplicated" 2>&1 | tail -3
echo "== working build"; timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "deduplicated" 2>&1 | tail -3
echo "== working build, GNA off"; DTP_NO_GNA_LNLIN=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "deduplicated" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "groupnorm or gn or statistics or layernorm_fold or resident" 2>&1 | tail -15
timeout 600 python tools/scratch/gn_large_mean.py > gpurun_out/r06_gn_large_mean_after.log 2>&1; grep -v amdgpu gpurun_out/r06_gn_large_mean_after.log | tail -20
rm -f gpurun_out/ab_summary.log
bash tools/ab.sh tools/ab/libdtp_r05.so b1 2
grep -h graph_nodes gpurun_out/ab_b1_ref_1.log gpurun_out/ab_b1_new_1.log | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['ms_per_step'], d['config']['graph_nodes'])"
