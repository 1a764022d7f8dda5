#!/bin/bash
# round 6, final build: direct same-box A/B against the round-5 library (shipped tune table on both arms), then what the driver does
rm -f gpurun_out/ab_summary.log
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
bash tools/ab.sh tools/ab/libdtp_r05.so all 3
cp gpurun_out/ab_summary.log gpurun_out/r06_ab_r05_vs_r06_final.txt
unset DTP_TUNE_CACHE
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | grep -v amdgpu.ids | tail -5
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_driver_style_bench.log 2> gpurun_out/r06_driver_style_bench.err ) 2>&1 | tail -4
tail -1 gpurun_out/r06_driver_style_bench.log | wc -c
tail -1 gpurun_out/r06_driver_style_bench.log | cut -c1-200
