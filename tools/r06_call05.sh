#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "groupnorm or gn or statistics or layernorm_fold or resident" 2>&1 | tail -5
bash tools/r06_dumpcmp.sh tools/ab/libdtp_r05.so 2 all
