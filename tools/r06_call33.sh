#!/bin/bash
# round 6, last commit: the op tests and the full-size tests (without the 4-minute oracle run, which the suite of r06_call24.sh covered) once more
( time timeout 700 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -x -k "not config1_and_config2" ) 2>&1 | tail -5
