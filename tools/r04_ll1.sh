#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "lnlin" 2>&1 | tail -5
tools/micro/exp_rate 2>&1 | tee gpurun_out/r04_exp_rate.log
cp tools/ab/tc_ws.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile > /dev/null 2>&1
for i in 1 2; do
DTP_NO_LNLIN=1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no lnlin at all', d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('with           ', d['ms_per_step'])"
done
grep ",ll " /tmp/ab_tc.txt
cp /tmp/ab_tc.txt gpurun_out/r04_tc_ll.txt
