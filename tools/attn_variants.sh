#!/bin/bash
# diagnostic builds of attention.hip with one ingredient of the key-tile loop removed (results are wrong on purpose): tools/ab/libdtp_attn_<x>.so
# run here (CPU container, hipcc cross-compiles); then  python tools/bench_attn.py  on the GPU times every variant it finds
set -e
cd "$(dirname "$0")/.."
OBJ=diffusiontexturepainting_amd/csrc/build
mkdir -p tools/ab
for v in ${ATTN_VARIANTS:-NO_MFMA NO_LDSREAD NO_EXP NO_STAGE}; do
  lc=$(echo $v | tr 'A-Z' 'a-z' | tr -d '_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fPIC -DDTP_ATTN_$v -c diffusiontexturepainting_amd/csrc/attention.hip -o /tmp/attn_$lc.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdtp_attn_$lc.so $(ls $OBJ/*.o | grep -v /attention.o) /tmp/attn_$lc.o
done
ls -la tools/ab/ | grep attn
