#!/bin/bash
# tools/r04_tall_rule.sh: the conv tile-height rule of round 4 (product) against the previous one (tools/_ab/lib_prev.so): sha, ABAB decode
# time at 8 / 4 / 1 songs, then the VAE + config tests on the product.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_tall_rule.txt
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_new.so; cp tools/_ab/lib_prev.so /tmp/_old.so
{
for B in ${BATCHES:-8 8 8 7 4 1}; do
  for v in old new; do
    cp /tmp/_$v.so $LIB
    echo "$v: $(VB=$B python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done
cp /tmp/_new.so $LIB
timeout 1200 python -m pytest tests/test_vae_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -2
} > $OUT 2>&1
cp /tmp/_new.so $LIB
cat $OUT
