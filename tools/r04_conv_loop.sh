#!/bin/bash
# tools/r04_conv_loop.sh: the conv main loop with the next chunk's first weight tile requested under the last tap (product, two-stage ring)
# (tools/r04_conv_ring.patch with CONV_NST_4WAVE=2; NOT KEPT)
# against the round-3 loop (tools/_ab/lib_oldloop.so: tap 0 of every chunk requested and waited for behind the window staging)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_conv_loop.txt
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_new.so; cp tools/_ab/lib_oldloop.so /tmp/_old.so
{
for r in 1 2 3 4; do
  for v in old new; do
    cp /tmp/_$v.so $LIB
    echo "$v loop: $(python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done
} > $OUT 2>&1
cp /tmp/_new.so $LIB
cat $OUT
