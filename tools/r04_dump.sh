#!/bin/bash
# per-launch table of one profiled batch-1 stamp (every launch bracketed by HIP events): gpurun_out/r04_launches_b1.csv
mkdir -p gpurun_out
timeout 900 python bench.py --no-cpu-baseline --no-extras --dump-launches gpurun_out/r04_launches_b1.csv > gpurun_out/r04_dump_b1.log 2>&1
tail -1 gpurun_out/r04_dump_b1.log | cut -c1-200
