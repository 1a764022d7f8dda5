#!/bin/bash
# Same-box A/B of two builds inside ONE gpurun call: the reference build ($1, default tools/ab/libdtp_head.so = a copy of the
# library built from HEAD before the change) against the working build, alternating, N runs per arm.
#   tools/ab.sh [ref.so] [what] [runs]      what: b1 (configs[1], default) | b8 (configs[2]) | 256 (256^2 / 20 steps) | all
# Extra environment for the working arm only: AB_ENV="DTP_FOO=1 DTP_BAR=0".  Logs: gpurun_out/ab_<what>_{ref,new}_<i>.log
REF=${1:-tools/ab/libdtp_head.so}; WHAT=${2:-b1}; RUNS=${3:-3}
mkdir -p gpurun_out
export DTP_TUNE_CACHE=${DTP_TUNE_CACHE:-/tmp/ab_tc.txt}
COMMON="--no-cpu-baseline --no-extras --no-profile"
run() {  # $1 = tag, $2.. = bench arguments
  local tag=$1; shift
  for i in $(seq 1 $RUNS); do
    DTP_LIB=$REF timeout 900 python bench.py $COMMON "$@" > gpurun_out/ab_${tag}_ref_$i.log 2>&1
    env $AB_ENV timeout 900 python bench.py $COMMON "$@" > gpurun_out/ab_${tag}_new_$i.log 2>&1
  done
  for arm in ref new; do
    printf "%s %s:" $tag $arm
    for i in $(seq 1 $RUNS); do python - gpurun_out/ab_${tag}_${arm}_$i.log <<'PY'
import json, sys
v = None
for ln in open(sys.argv[1]):
    if ln.startswith('{'):
        try: v = json.loads(ln)
        except Exception: pass
print(" %.2f ms" % v["ms_per_step"] if v else " FAILED", end="")
PY
    done; echo
  done
}
case $WHAT in
  b1) run b1 ;;
  b8) run b8 --batch 8 --steps 3 --warmup 1 ;;
  256) run 256 --res 256 ;;
  all) run b1; run 256 --res 256; run b8 --batch 8 --steps 3 --warmup 1 ;;
  *) echo "unknown arm $WHAT"; exit 2 ;;
esac 2>&1 | tee -a gpurun_out/ab_summary.log
