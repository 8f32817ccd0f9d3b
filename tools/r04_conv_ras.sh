#!/bin/bash
# tools/r04_conv_ras.sh: the XCD-aware rasterisation of the conv grid (ACE355_CONV_RAS, conv.hip) against the plain (m, n, b) grid:
# bit-identity of the decode / encode (sha of the waveforms), ABAB decode time, per-launch durations under the kernel tracer.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r04_conv_ras.txt
mkdir -p gpurun_out
{
echo "== ABAB, tools/vae_ab_check.py (8 x 30 s decode, mean of 5; then a 2 x 10 s encode)"
for r in 1 2 3; do
  for v in 0 1; do
    echo "ACE355_CONV_RAS=$v: $(ACE355_CONV_RAS=$v python tools/vae_ab_check.py 2>&1 | tr '\n' ' ')"
  done
done
for v in 0 1; do
  echo "== per-launch durations of one decode, ACE355_CONV_RAS=$v (rocprofv3 --kernel-trace)"
  rm -rf /tmp/ras_$v
  ACE355_CONV_RAS=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/ras_$v -- python tools/vae_trace.py > /dev/null 2>&1
  python tools/vae_trace_list.py /tmp/ras_$v
done
} > $OUT 2>&1
timeout 900 python -m pytest tests/test_vae_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $OUT
cat $OUT
