#!/bin/bash
# tools/r04_episnake.sh: producer-side Snake (ACE355_VAE_EPISNAKE, vae.hip / conv.hip ConvArgs::osnake_a) against every Snake in its
# reader's window staging: the VAE parity tests under both settings, ABAB decode time, per-launch durations under the kernel tracer.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04_episnake.txt
mkdir -p gpurun_out
{
for v in 1 0; do
  echo "== tests/test_vae_gpu.py with ACE355_VAE_EPISNAKE=$v"
  ACE355_VAE_EPISNAKE=$v timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -x -q -s 2>&1 | grep -E "^vae |SNR|passed|failed|Error" | cut -c1-260
done
echo "== ABAB, tools/vae_ab_check.py (8 x 30 s decode, mean of 5; then a 2 x 10 s encode)"
for r in 1 2 3; do
  for v in 0 1; do
    echo "ACE355_VAE_EPISNAKE=$v: $(ACE355_VAE_EPISNAKE=$v python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')"
  done
done
for v in 0 1; do
  echo "== per-launch durations of one decode, ACE355_VAE_EPISNAKE=$v (rocprofv3 --kernel-trace)"
  rm -rf /tmp/es_$v
  ACE355_VAE_EPISNAKE=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/es_$v -- python tools/vae_trace.py > /dev/null 2>&1
  python tools/vae_trace_list.py /tmp/es_$v
done
} > $OUT 2>&1
cat $OUT
