#!/bin/bash
# 256^2 / 20 steps: the GroupNorm-on-load proj_in on / off, same library, interleaved
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
for i in 1 2 3; do for arm in 0 1; do
  DTP_NO_GNA_LNLIN=$arm timeout 600 python bench.py --res 256 --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('256 no_gna=$arm', d['ms_per_step'], d['config']['graph_nodes'])"
done; done
for i in 1 2; do for arm in 0 1; do
  DTP_NO_GNA_LNLIN=$arm timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('b1 no_gna=$arm', d['ms_per_step'], d['config']['graph_nodes'])"
done; done
