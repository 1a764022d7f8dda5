#!/bin/bash
# after the tuner rule (engine.hip: no kernel changed, but the PMC summaries are guarded by the hash of ALL csrc sources): the tuner path
# exercised by tests that tune, then the two PMC summaries again
mkdir -p gpurun_out
export DTP_ROUND=r05
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/tc.txt
export DTP_TUNE_CACHE=/tmp/tc.txt
timeout 400 python -m pytest tests/test_gpu_round2.py -q -x -k "tune or destroy" > gpurun_out/r05_tuner_tests.log 2>&1; tail -2 gpurun_out/r05_tuner_tests.log
DTP_TUNE_CACHE=/tmp/tc_fresh.txt DTP_TUNE_SEED=/dev/null timeout 500 python bench.py --res 128 --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-profile > gpurun_out/r05_tuner_fresh128.log 2>&1; tail -1 gpurun_out/r05_tuner_fresh128.log | cut -c1-160; grep -c "ws2 5[34] 1" /tmp/tc_fresh.txt; grep -c "ws2 1[2-5] 1" /tmp/tc_fresh.txt
bash tools/pmc_unet.sh
cp gpurun_out/r05_pmc_unet_traffic.json profiles/r05_pmc_unet_traffic.json
bash tools/pmc_unet_mfma.sh > gpurun_out/r05_pmc_unet_mfma.log 2>&1
cp gpurun_out/r05_pmc_unet_mfma.json profiles/r05_pmc_unet_mfma.json
grep -o '"kernel_source_hash": "[0-9a-f]*"' gpurun_out/r05_pmc_unet_traffic.json gpurun_out/r05_pmc_unet_mfma.json
