#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "lnlin or layernorm_fold" > gpurun_out/r03_ops22.log 2>&1
timeout 400 python tools/diag_lnlin.py > gpurun_out/r03_diag_lnlin22.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so timeout 400 python tools/diag_lnlin.py > gpurun_out/r03_diag_lnlin22_head.log 2>&1
