#!/bin/bash
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x > gpurun_out/r05_poly_ops.log 2>&1; tail -3 gpurun_out/r05_poly_ops.log
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_round2.py -q -x > gpurun_out/r05_poly_engine.log 2>&1; tail -3 gpurun_out/r05_poly_engine.log
grep -h "max abs\|err" gpurun_out/r05_poly_engine.log | head -5
echo "== lnlin: poly GELU (shipped) vs A&S + v_rcp (head)"
timeout 300 python tools/bench_lnlin.py --child 2>&1 | grep -v amdgpu.ids
DTP_LIB=tools/ab/libdtp_head.so timeout 300 python tools/bench_lnlin.py --child 2>&1 | grep -v amdgpu.ids
echo "== ref = poly GELU only (GroupNorm load order is the difference)"
bash tools/ab.sh tools/ab/libdtp_poly.so all 2
echo "== ref = head (poly GELU + GroupNorm load order are the difference)"
bash tools/ab.sh tools/ab/libdtp_head.so all 2
