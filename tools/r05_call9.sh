#!/bin/bash
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
python tools/scratch/lnlin_occ.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x > gpurun_out/r05_rcp_ops.log 2>&1; tail -3 gpurun_out/r05_rcp_ops.log
timeout 600 python -m pytest tests/test_gpu_engine.py -q -x > gpurun_out/r05_rcp_engine.log 2>&1; tail -3 gpurun_out/r05_rcp_engine.log
echo "== lnlin, fast rcp (shipped) vs IEEE division"
timeout 300 python tools/bench_lnlin.py --child 2>&1 | grep -v amdgpu.ids
DTP_LIB=tools/ab/libdtp_ieeediv.so timeout 300 python tools/bench_lnlin.py --child 2>&1 | grep -v amdgpu.ids
bash tools/ab.sh tools/ab/libdtp_ieeediv.so all 2
