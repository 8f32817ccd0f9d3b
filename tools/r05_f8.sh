#!/bin/bash
# tools/r05_f8.sh: the fused residual unit on the 8-wave tile with 32 x 128 wave shapes (conv_kernel<128, 256, 1>, ACE355_CONV_F8) against the
# 4-wave form: waveform sha (the two forms must agree bit for bit), ABAB decode times at 8 / 1 songs, parity tests, clock probe, trace.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_f8.txt
mkdir -p gpurun_out
{
echo "== ABAB by ACE355_CONV_F8 (0 = 4-wave fused form, 1 = 8-wave): 8 x 30 s decode"
for r in 1 2 3; do for v in 1 0; do
  echo "F8=$v: $(ACE355_CONV_F8=$v python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
done; done
echo "== one song"
for v in 1 0 1 0; do echo "F8=$v: $(ACE355_CONV_F8=$v VB=1 python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | head -1)"; done
echo "== default rule: $(python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
if [ "$1" != "notests" ]; then
echo "== parity tests (default rule)"
timeout 1500 python -m pytest tests/test_vae_gpu.py tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -5
echo "== parity tests, ACE355_CONV_F8=1 (8-wave form at every size)"
ACE355_CONV_F8=1 timeout 1500 python -m pytest tests/test_vae_gpu.py -m gpu -x -q 2>&1 | tail -5
fi
for v in 1 0; do
  echo "== clock probe, ACE355_CONV_F8=$v"
  ACE355_CONV_F8=$v ACE355_CONV_CLK=1 python tools/vae_trace.py 2>&1 | grep "conv clk" | grep "Cin=128 N=128" | tail -6
done
echo "== per-launch durations, default rule"
rm -rf /tmp/ct_A
rocprofv3 --kernel-trace --output-format csv -d /tmp/ct_A -- python tools/vae_trace.py > /dev/null 2>&1
python tools/vae_trace_list.py /tmp/ct_A | tail -34
} > $OUT 2>&1
cat $OUT
