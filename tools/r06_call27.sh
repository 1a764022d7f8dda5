#!/bin/bash
# HIP runtime knobs never tried before: where kernel arguments live, graph nodes as pre-built AQL packets, the fence scope between kernels
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
export DTP_TUNE_CACHE=/tmp/ab_tc.txt
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"; }
run() {  # $1 = label, $2 = env assignment or "X=1" dummy
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | line "b1  $1"
  env $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile --res 256 2>/dev/null | line "256 $1"
}
for pass in 1 2; do
  run default DTP_DUMMY=1
  run dev_kernarg=1 HIP_FORCE_DEV_KERNARG=1
  run dev_kernarg=0 HIP_FORCE_DEV_KERNARG=0
  run packet_capture=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run packet_capture=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run opt_flush=0 AMD_OPT_FLUSH=0
  run opt_flush=1 AMD_OPT_FLUSH=1
done 2>&1 | tee gpurun_out/r06_runtime_knobs.txt
