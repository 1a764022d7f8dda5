#!/bin/bash
DTP_SKIP_FULLSIZE=1 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -q -x -k "not batch16 and not trained_like and not fp8" 2>&1 | tail -5
bash tools/r06_dumpcmp.sh tools/ab/libdtp_r05.so 2 all
