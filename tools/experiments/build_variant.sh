#!/bin/bash
# A/B builds of the C-ABI library with ONE source compiled under different -D switches, selected at
# run time through LANCZOSNET_HIP_LIB (the library carries the in-tree SONAME, so the torch
# extension binds to it as well).
#   tools/experiments/build_variant.sh conv_forward16.hip phases:"-DLNZ_F16_PHASES" ...
#   -> tools/experiments/_variants/liblnz_<source stem>_<name>.so
set -e
cd "$(dirname "$0")/../../lanczosnet_amd/csrc"
SRC=$1; shift
STEM=${SRC%.hip}
OUT=../../tools/experiments/_variants
mkdir -p $OUT
OBJS=$(ls *.o | grep -v "^$STEM.o$")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed $flags -c $SRC -o $OUT/${STEM}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,liblanczosnet_hip.so -o $OUT/liblnz_${STEM}_$name.so $OBJS $OUT/${STEM}_$name.o
  rm -f $OUT/${STEM}_$name.o
  echo built $name "($flags)"
done
