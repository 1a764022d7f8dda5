#!/bin/bash
# A/B reference build of the WORKING tree with extra compiler flags: tools/ab/libdtp_<name>.so   (run here, hipcc cross-compiles)
#   tools/ab_build.sh ieeediv -DDTP_IEEE_DIV          then on the GPU:  tools/ab.sh tools/ab/libdtp_ieeediv.so all 2
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=/tmp/ab_build_$name; mkdir -p $out tools/ab
for f in diffusiontexturepainting_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  case $b in attention|attn_dma) extra="-ffast-math";; esac
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -Iinclude -c $f -o $out/$b.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libdtp_$name.so $out/*.o
ls -la tools/ab/libdtp_$name.so
