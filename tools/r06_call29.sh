#!/bin/bash
# (run while the package applied the setting by default; it has been an opt-in since: DTP_RUNTIME_ENV=1)
# the HIP runtime configuration applied by the package itself (diffusiontexturepainting_amd/__init__.py) against the shell-set variable and against
# $DTP_RUNTIME_ENV=0 (= the runtime's default, packet capture on); then the GPU suite and the driver-style run with it
mkdir -p gpurun_out
cp diffusiontexturepainting_amd/tune_seed.txt /tmp/ab_tc.txt
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'])"; }
run() {
  env DTP_TUNE_CACHE=/tmp/ab_tc.txt $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile 2>/dev/null | line "b1  $1"
  env DTP_TUNE_CACHE=/tmp/ab_tc.txt $2 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-profile --res 256 2>/dev/null | line "256 $1"
}
for i in 1 2; do
  run "DTP_RUNTIME_ENV=1 (the package sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 itself)" DTP_RUNTIME_ENV=1
  run "DTP_RUNTIME_ENV=0 (runtime default)" DTP_RUNTIME_ENV=0
  run "shell DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done 2>&1 | tee gpurun_out/r06_runtime_env_in_package.txt
( time DTP_RUNTIME_ENV=1 timeout 1500 python -m pytest tests -q -m gpu --durations=8 -x ) > gpurun_out/r06_gpu_suite.log 2>&1
tail -3 gpurun_out/r06_gpu_suite.log
DTP_RUNTIME_ENV=1 DTP_ROUND=r06 bash tools/driver_style_check.sh 2>&1 | tail -8
