#!/bin/bash
# HIP runtime knobs, second pass: around DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (graph nodes through the ordinary dispatch path: -1.0 % / -2.4 % in r06_call27.sh)
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"; }
run() {
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | line "b1  $1"
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile --res 256 2>/dev/null | line "256 $1"
}
{
run default DTP_DUMMY=1
run pc=0 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run pc=0,dev_kernarg=1 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HIP_FORCE_DEV_KERNARG=1"
run pc=0,dev_kernarg=0 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HIP_FORCE_DEV_KERNARG=0"
run pc=0,kernarg_copy_opt=0 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_HIP_KERNARG_COPY_OPT=0"
run pc=0,max_batch=64 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_MAX_BATCH_SIZE=64"
run pc=0,max_batch=8192 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_MAX_BATCH_SIZE=8192"
run pc=0,hdp_wa=0 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0"
run pc=1,graph_batch=1 "DEBUG_HIP_GRAPH_BATCH_SIZE=1"
run pc=1,graph_batch=4096 "DEBUG_HIP_GRAPH_BATCH_SIZE=4096"
run pc=0 "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run default DTP_DUMMY=1
for i in 1 2; do
  timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile --batch 8 --steps 3 --warmup 1 2>/dev/null | line "b8  default"
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 900 python bench.py --no-cpu-baseline --no-extras --no-profile --batch 8 --steps 3 --warmup 1 2>/dev/null | line "b8  pc=0"
done
} 2>&1 | tee gpurun_out/r06_runtime_knobs2.txt
