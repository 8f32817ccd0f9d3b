#!/bin/bash
# tools/r05_variants_vae.sh NAME...: the product library (A) against csrc/_variants/libace355_NAME.so on the 8 x 30 s decode, interleaved, 2 rounds
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_A.so
for r in 1 2 3; do
  cp /tmp/_A.so $LIB; echo "A: $(python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | head -1)"
  for n in "$@"; do
    cp ace-step-1.5-for-windows_amd/csrc/_variants/libace355_$n.so $LIB; echo "$n: $(python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | head -1)"
  done
done
cp /tmp/_A.so $LIB
