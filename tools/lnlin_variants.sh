#!/bin/bash
# diagnostic builds of lnlin.hip with one ingredient of the unit loop removed (results are wrong on purpose): tools/ab/libdtp_lnlin_<x>.so
# run here (CPU container, hipcc cross-compiles); then  python tools/bench_lnlin.py  on the GPU times every variant it finds
set -e
cd "$(dirname "$0")/.."
OBJ=diffusiontexturepainting_amd/csrc/build
mkdir -p tools/ab
for v in ${LNLIN_VARIANTS:-NO_MFMA NO_LDSREAD NO_GELU}; do
  lc=$(echo $v | tr 'A-Z' 'a-z' | tr -d '_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDTP_LNLIN_$v -c diffusiontexturepainting_amd/csrc/lnlin.hip -o /tmp/lnlin_$lc.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdtp_lnlin_$lc.so $(ls $OBJ/*.o | grep -v /lnlin.o) /tmp/lnlin_$lc.o
done
ls -la tools/ab/ | grep lnlin
