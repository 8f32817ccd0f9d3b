#!/bin/bash
# tools/r04_episnake_bench.sh: the Snake placement switch through bench.py itself (same box, ABAB): ms per 8-song pass and the VAE part
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_episnake_bench_ab.txt
{
for r in 1 2 3; do
  for v in 0 1; do
    ACE355_VAE_EPISNAKE=$v python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); r = d['roofline']
print('ACE355_VAE_EPISNAKE=$v', round(d['ms_per_step'], 2), 'ms per pass,', round(d['value'], 3), 'songs/s; vae', round(r['vae_conv_ms_per_pass'], 2), 'ms', round(r['vae_conv_tflops']), 'TF/s; gemm', round(r['gemm_ms_per_pass'], 1), 'attn', round(r['attn_ms_per_pass'], 1))"
  done
done
} > $OUT 2>&1
cat $OUT
