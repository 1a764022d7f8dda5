#!/bin/bash
# attention max3 chain: op parity; how long the one-rank RCCL bench takes and whether MSCCL / debug settings change it
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "attention" > gpurun_out/r03_ops19.log 2>&1
timeout 300 python tools/diag_attn.py > gpurun_out/r03_diag_attn19.log 2>&1
( time DTP_BENCH_FORCE_DIST=1 NCCL_DEBUG=INFO timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --res 64 --ddim-steps 4 --batch 2 --no-cpu-baseline --no-extras --no-profile ) > gpurun_out/r03_rccl_a.log 2>&1
( time DTP_BENCH_FORCE_DIST=1 RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0 timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --res 64 --ddim-steps 4 --batch 2 --no-cpu-baseline --no-extras --no-profile ) > gpurun_out/r03_rccl_b.log 2>&1
( time timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -k "tune_table" ) > gpurun_out/r03_tune19.log 2>&1
