#!/bin/bash
# tools/r04_conv_ring.sh: three-stage weight ring of the 4-wave conv tiles (product) against the two-stage build (tools/_ab/lib_nst2.so,
# -DCONV_NST_4WAVE=2).  NOT KEPT: the loop lives in tools/r04_conv_ring.patch (apply it to csrc/conv.hip to re-run).  Checks: bit-identity of decode / encode (sha), ABAB decode time, per-launch durations, the VAE tests on the product.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04_conv_ring.txt
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_ring3.so; cp tools/_ab/lib_nst2.so /tmp/_ring2.so
{
echo "== ABAB, tools/vae_ab_check.py (8 x 30 s decode, mean of 5; then a 2 x 10 s encode)"
for r in 1 2 3; do
  for v in 2 3; do
    cp /tmp/_ring$v.so $LIB
    echo "ring of $v stages: $(python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done
for v in 2 3; do
  cp /tmp/_ring$v.so $LIB
  echo "== per-launch durations of one decode, ring of $v stages (rocprofv3 --kernel-trace)"
  rm -rf /tmp/ring_$v
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ring_$v -- python tools/vae_trace.py > /dev/null 2>&1
  python tools/vae_trace_list.py /tmp/ring_$v
done
cp /tmp/_ring3.so $LIB
echo "== tests/test_vae_gpu.py + test_configs_gpu.py on the product"
timeout 1200 python -m pytest tests/test_vae_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -3
} > $OUT 2>&1
cp /tmp/_ring3.so $LIB
cat $OUT | grep -v "us grid"
