#!/bin/bash
# reduce-in-concat-GN: op tests, B = 16 extra config (VAE programs in sub-batches), and the A/B (the "new" arm is the switch-OFF arm)
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "reduce_groupnorm or groupnorm or vae" > gpurun_out/r05_rgn_tests.log 2>&1
tail -4 gpurun_out/r05_rgn_tests.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -q -x > gpurun_out/r05_rgn_tests2.log 2>&1
tail -4 gpurun_out/r05_rgn_tests2.log
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/r05_b16.log 2> gpurun_out/r05_b16.err
tail -3 gpurun_out/r05_b16.err
tail -1 gpurun_out/r05_b16.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['config'].get('graph_nodes'))
for k,v in d['extra_configs'].items(): print(k, v)
print(d['roofline'])
"
AB_ENV="DTP_NO_REDUCE_IN_CONCAT_GN=1" bash tools/ab.sh diffusiontexturepainting_amd/libdtp.so all 2
cp /tmp/ab_tc.txt gpurun_out/r05_tune_seed_call6.txt
