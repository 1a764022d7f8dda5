#!/bin/bash
# round 6, call 2: the GroupNorm large-mean diagnostic + the op tests this commit touched
mkdir -p gpurun_out
timeout 600 python tools/scratch/gn_large_mean.py > gpurun_out/r06_gn_large_mean.log 2>&1; tail -25 gpurun_out/r06_gn_large_mean.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or xattn or cross_attention or reduce_groupnorm or gemm_dense or conv3x3" --durations=5 2>&1 | tail -15
