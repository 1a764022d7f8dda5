#!/bin/bash
# diagnostic builds of lnlin.hip with one ingredient of the unit loop removed (results are wrong on purpose): tools/ab/libdtp_lnlin_<x>.so
# run here (CPU container, hipcc cross-compiles); then  python tools/bench_lnlin.py  on the GPU times every variant it finds
set -e
cd "$(dirname "$0")/.."
OBJ=diffusiontexturepainting_amd/csrc/build
mkdir -p tools/ab
for v in ${LNLIN_VARIANTS:-NO_MFMA NO_LDSREAD NO_GELU NO_DMA NO_BARRIER NO_EPI NO_STORE NO_EPI+NO_MFMA NO_EPI+NO_DMA}; do
  lc=$(echo $v | tr 'A-Z' 'a-z' | tr -d '_' | tr '+' '_')
  defs=$(echo $v | tr '+' '\n' | sed 's/^/-DDTP_LNLIN_/' | tr '\n' ' ')   # A+B = both ingredients removed
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $defs -c diffusiontexturepainting_amd/csrc/lnlin.hip -o /tmp/lnlin_$lc.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdtp_lnlin_$lc.so $(ls $OBJ/*.o | grep -v /lnlin.o) /tmp/lnlin_$lc.o
done
ls -la tools/ab/ | grep lnlin
