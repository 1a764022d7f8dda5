#!/bin/bash
# diagnostic builds of conv_ws.hip with one ingredient of the main loop removed (results are wrong on purpose): tools/ab/libdtp_ws_<x>.so
# run here (CPU container, hipcc cross-compiles); then e.g.  DTP_LIB=tools/ab/libdtp_ws_nomfma.so python tools/diag_ws.py --cold --ws --noreduce
set -e
cd "$(dirname "$0")/.."
OBJ=diffusiontexturepainting_amd/csrc/build
mkdir -p tools/ab
for v in ${WS_VARIANTS:-NO_MFMA NO_LDSREAD NO_WLOAD NO_DMA}; do
  lc=$(echo $v | tr 'A-Z' 'a-z' | tr -d '_')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDTP_WS_$v -c diffusiontexturepainting_amd/csrc/conv_ws.hip -o /tmp/conv_ws_$lc.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdtp_ws_$lc.so $(ls $OBJ/*.o | grep -v conv_ws.o) /tmp/conv_ws_$lc.o
done
ls -la tools/ab/
