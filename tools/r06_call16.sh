#!/bin/bash
# which launches of a 256^2 stamp differ between the round-5 library and the working build?
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
DTP_LIB=tools/ab/libdtp_r05.so timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r06_dump_ref_256.csv > /dev/null 2>&1
timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r06_dump_new_256.csv > /dev/null 2>&1
python tools/dump_cmp.py gpurun_out/r06_dump_ref_256.csv gpurun_out/r06_dump_new_256.csv 400 > gpurun_out/r06_dumpcmp_256.txt
python - <<'P'
import re
for ln in open("gpurun_out/r06_dumpcmp_256.txt"):
    m = re.match(r"\s*([\d.]+)\s+([\d.]+)", ln)
    if not m: print(ln.rstrip()); continue
    a, b = float(m.group(1)), float(m.group(2))
    if abs(a - b) > 25.0 or a == 0 or b == 0: print(ln.rstrip())
P
