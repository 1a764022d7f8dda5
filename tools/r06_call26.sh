#!/bin/bash
# round 6: the final A/B once more on another box, three runs per arm (round-5 library rebuilt from git vs the final build, shipped tune table on both)
mkdir -p gpurun_out; rm -f gpurun_out/ab_summary.log
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
DTP_TUNE_CACHE=/tmp/ab_tc.txt bash tools/ab.sh tools/ab/libdtp_r05.so all 3 > /dev/null 2>&1
cp gpurun_out/ab_summary.log gpurun_out/r06_ab_r05_vs_r06_final_box2.txt; cat gpurun_out/r06_ab_r05_vs_r06_final_box2.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('final build on this box:', d['ms_per_step'], 'ms,', d['roofline']['peak_measured'], 'TFLOP/s bare MFMA')" | tee -a gpurun_out/r06_ab_r05_vs_r06_final_box2.txt
