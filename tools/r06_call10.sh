#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "register_chained" 2>&1 | tail -5
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
for i in 1 2; do for arm in 0 1; do
  DTP_FFCHAIN=$arm timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile --batch 8 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('b8 ffchain=$arm', d['ms_per_step'], d['config']['graph_nodes'])"
done; done
DTP_FFCHAIN=1 timeout 900 python bench.py --no-cpu-baseline --no-extras --batch 8 --steps 2 --warmup 1 --dump-launches gpurun_out/r06_dump_b8_ffchain.csv > /dev/null 2>&1
DTP_FFCHAIN=0 timeout 900 python bench.py --no-cpu-baseline --no-extras --batch 8 --steps 2 --warmup 1 --dump-launches gpurun_out/r06_dump_b8_noffchain.csv > /dev/null 2>&1
python tools/dump_cmp.py gpurun_out/r06_dump_b8_noffchain.csv gpurun_out/r06_dump_b8_ffchain.csv 400 | python -c "
import sys,re
for ln in sys.stdin:
    m=re.match(r'\s*([\d.]+)\s+([\d.]+)',ln)
    if not m: print(ln.rstrip()); continue
    a,b=float(m.group(1)),float(m.group(2))
    if abs(a-b)>300 or a==0 or b==0: print(ln.rstrip())"
