#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "register_chained" 2>&1 | tail -25
