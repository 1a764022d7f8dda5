#!/bin/bash
# per-launch tables of one profiled stamp for the reference build ($1, default tools/ab/libdtp_head.so) and the working build, same box
REF=${1:-tools/ab/libdtp_head.so}
mkdir -p gpurun_out
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
DTP_LIB=$REF timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r05_dump_ref_b1.csv > gpurun_out/r05_dump_ref_b1.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r05_dump_new_b1.csv > gpurun_out/r05_dump_new_b1.log 2>&1
python tools/dump_cmp.py gpurun_out/r05_dump_ref_b1.csv gpurun_out/r05_dump_new_b1.csv 40
DTP_LIB=$REF timeout 900 python bench.py --res 256 --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r05_dump_ref_256.csv > gpurun_out/r05_dump_ref_256.log 2>&1
timeout 900 python bench.py --res 256 --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r05_dump_new_256.csv > gpurun_out/r05_dump_new_256.log 2>&1
python tools/dump_cmp.py gpurun_out/r05_dump_ref_256.csv gpurun_out/r05_dump_new_256.csv 30
