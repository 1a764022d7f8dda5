#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "register_chained" 2>&1 | tail -15
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
for arm in 0 1; do for i in 1 2; do
  DTP_FFCHAIN=$arm timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile --batch 8 --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('b8 ffchain=$arm', d['ms_per_step'], d['config']['graph_nodes'])"
done; done
for arm in 0 1; do
  DTP_FFCHAIN=$arm timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('b1 ffchain=$arm', d['ms_per_step'], d['config']['graph_nodes'])"
done
