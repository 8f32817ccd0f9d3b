#!/bin/bash
# tools/r05_conv_v2.sh: version 2 of conv_kernel's chunk / tap loop (conv.hip: ACE355_CONV_V2) against version 1 (csrc/_variants/libace355_convv1.so,
# tools/build_variant.sh convv1 conv.hip -DACE355_CONV_V2=0): the VAE / conv parity tests on the new library, same-box ABAB of the 8 x 30 s decode
# (tools/vae_ab_check.py prints the waveform sha: the two versions must agree bit for bit), one song, and the in-kernel clock probe of both.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_conv_v2.txt
mkdir -p gpurun_out
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
OLD=ace-step-1.5-for-windows_amd/csrc/_variants/libace355_convv1.so
cp $LIB /tmp/_A.so; cp $OLD /tmp/_B.so
{
echo "== parity tests on version 2"
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -5
echo "== ABAB (A = version 2, B = version 1): 8 x 30 s decode, mean of 5; 2 x 10 s encode sha"
for r in 1 2 3; do
  for v in A B; do
    cp /tmp/_$v.so $LIB
    echo "$v: $(python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done
echo "== one song"
for v in A B A B; do
  cp /tmp/_$v.so $LIB
  echo "$v: $(VB=1 python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | head -1)"
done
for v in A B; do
  cp /tmp/_$v.so $LIB
  echo "== clock probe of one interior workgroup per launch, $v (8 x 30 s decode)"
  ACE355_CONV_CLK=1 python tools/vae_trace.py 2>&1 | grep "conv clk" | head -60
done
cp /tmp/_A.so $LIB
} > $OUT 2>&1
cat $OUT
