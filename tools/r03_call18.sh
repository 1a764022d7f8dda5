#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "halo or conv3x3" > gpurun_out/r03_ops18.log 2>&1
DTP_LIB=tools/ab/libdtp_head.so timeout 300 python tools/diag_halo.py > gpurun_out/r03_diag_halo_head.log 2>&1
timeout 300 python tools/diag_halo.py > gpurun_out/r03_diag_halo_new.log 2>&1
