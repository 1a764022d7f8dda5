#!/bin/bash
# the grid-synchronised GroupNorm (norm.hip gn_grid_kernel): parity + bit-identity with the two-launch form, the stamp-level bit-identity /
# oracle checks that sit on top of it, then a same-box A/B by environment switch ($DTP_NO_GN_GRID=1 = the two launches)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "groupnorm" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -q -x -k "dedup or bit_identical" 2>&1 | tail -4
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['config']['graph_nodes'])"; }
for i in 1 2; do for arm in 1 0; do
  DTP_NO_GN_GRID=$arm timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | line "b1 two_launch=$arm"
  DTP_NO_GN_GRID=$arm timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile --res 256 2>/dev/null | line "256 two_launch=$arm"
done; done 2>&1 | tee gpurun_out/r06_ab_gn_grid.txt
DTP_NO_GN_GRID=0 timeout 900 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --dump-launches gpurun_out/r06_dump_b1_gngrid.csv > /dev/null 2>&1
python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open('gpurun_out/r06_dump_b1_gngrid.csv')) if r['kind'] == '13']
for r in rows: print(r['label'], r['launches'], r['avg_us'])
PY
