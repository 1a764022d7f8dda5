#!/bin/bash
# eager launches (no hipGraph) against graph replay, with captured packets (runtime default) and without: device time per stamp and host time per stamp
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"; }
run() {
  env DTP_TUNE_CACHE=/tmp/ab_tc.txt $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | line "b1  $1"
  env DTP_TUNE_CACHE=/tmp/ab_tc.txt $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile --res 256 2>/dev/null | line "256 $1"
  env $2 timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -x -s -k "enqueue" 2>&1 | grep "host enqueue" | sed "s/^/512 $1: /"
}
{
run "graph, captured packets (runtime default)" "DTP_RUNTIME_ENV=0"
run "graph, packet capture off" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "eager launches" "DTP_RUNTIME_ENV=0 DTP_NO_GRAPH=1"
run "graph, captured packets (runtime default)" "DTP_RUNTIME_ENV=0"
run "eager launches" "DTP_RUNTIME_ENV=0 DTP_NO_GRAPH=1"
} 2>&1 | tee gpurun_out/r06_graph_vs_eager.txt
