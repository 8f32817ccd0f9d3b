#!/bin/bash
# tools/r05_conv_ab.sh VARIANT.so TAG [notests]: the product library (A) against another build of it (B, tools/build_variant.sh) on the VAE:
# parity tests on A, same-box ABAB of the 8 x 30 s decode and of one song (tools/vae_ab_check.py prints the waveform / latent sha),
# the in-kernel clock probe of the C = 128 launches of both, per-launch durations of A's decode under the kernel tracer.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OTHER=$1; TAG=$2
OUT=gpurun_out/r05_conv_ab_$TAG.txt
mkdir -p gpurun_out
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_A.so; cp $OTHER /tmp/_B.so
{
if [ "$3" != "notests" ]; then
echo "== parity tests on A"
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -5
fi
echo "== ABAB (A = product, B = $OTHER): 8 x 30 s decode, mean of 5; 2 x 10 s encode sha"
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
  echo "== clock probe of one interior workgroup per launch, $v (8 x 30 s decode; second decode)"
  ACE355_CONV_CLK=1 python tools/vae_trace.py 2>&1 | grep "conv clk" | tail -37
done
cp /tmp/_A.so $LIB
echo "== per-launch durations of A's decode (rocprofv3 --kernel-trace)"
rm -rf /tmp/ct_A
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct_A -- python tools/vae_trace.py > /dev/null 2>&1
python tools/vae_trace_list.py /tmp/ct_A
} > $OUT 2>&1
cat $OUT
