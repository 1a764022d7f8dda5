#!/bin/bash
# diagnostic builds of the attention kernels with one ingredient of the key-tile loop removed (results are wrong on purpose):
#   tools/ab/libdtp_attn_<x>.so;  ATTN_SRC=attn_dma (default, round 5: -DDTP_AD_<x>) or attention (round 4: -DDTP_ATTN_<x>)
# a variant name may combine switches with '+':  ATTN_VARIANTS="NO_MFMA NO_MFMA+NO_EXP TRACE"
# run here (CPU container, hipcc cross-compiles); then  python tools/bench_attn.py  on the GPU times every variant it finds
set -e
cd "$(dirname "$0")/.."
OBJ=diffusiontexturepainting_amd/csrc/build
SRC=${ATTN_SRC:-attn_dma}
if [ "$SRC" = attn_dma ]; then PFX=DTP_AD_; DEF="NO_MFMA NO_KREAD NO_VREAD NO_EXP NO_DMA"; else PFX=DTP_ATTN_; DEF="NO_MFMA NO_LDSREAD NO_EXP NO_STAGE"; fi
mkdir -p tools/ab
build_one() {
  v=$1
  lc=$(echo $v | tr 'A-Z' 'a-z' | tr -d '_' | tr '+' '-')
  defs=""
  for d in $(echo $v | tr '+' ' '); do defs="$defs -D$PFX$d"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fPIC $defs -c diffusiontexturepainting_amd/csrc/$SRC.hip -o /tmp/attn_$lc.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdtp_attn_$lc.so $(ls $OBJ/*.o | grep -v /$SRC.o) /tmp/attn_$lc.o
}
for v in ${ATTN_VARIANTS:-$DEF}; do build_one $v & done
wait
ls -la tools/ab/ | grep attn
