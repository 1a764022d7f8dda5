#!/bin/bash
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
python tools/scratch/gn_batch_indep.py 2>&1 | grep -v amdgpu.ids | grep -c "equal to batch-1 result False"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k "test_deduplicated_prefix_is_bit_identical or properties" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "norm or gn" 2>&1 | tail -2
