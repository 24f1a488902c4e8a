#!/bin/bash
# A/B builds of lnz_f32_linear: liblanczosnet_hip.so with f32_linear.hip compiled under different
# -D switches, selected at run time through LANCZOSNET_HIP_LIB (lanczosnet_amd/_lib.py loads it first;
# it carries the in-tree library's SONAME, so the torch extension binds to it as well).
#   tools/experiments/build_f32_variants.sh name1:"-DFLAGS" name2:"-DFLAGS" ...
set -e
cd "$(dirname "$0")/../../lanczosnet_amd/csrc"
OUT=../../tools/experiments/_variants
mkdir -p $OUT
OBJS=$(ls *.o | grep -v '^f32_linear.o$')
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed $flags -c f32_linear.hip -o $OUT/f32_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,liblanczosnet_hip.so -o $OUT/liblnz_f32_$name.so $OBJS $OUT/f32_$name.o
  rm -f $OUT/f32_$name.o
  echo built $name "($flags)"
done
