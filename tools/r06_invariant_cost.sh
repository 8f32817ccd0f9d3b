#!/bin/bash
# Round 6: what the launch-shape-independent mode (NativeHandler.shape_independent(): K rotation / split-K / split-KV / key-split attention off, one
# sampler chain) costs per request, DiT + decode, same box, interleaved
set -u
cd "$(dirname "$0")/.."
line() { python bench.py --no-cpu-baseline --no-roofline --batch $1 --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"; }
for rep in 1 2; do
  for b in 1 2 4 8; do
    echo "b=$b default                 $(line $b)"
    echo "b=$b shape-independent       $(ACE355_GEMM_KROT=0 ACE355_DUAL=0 line $b)"
  done
done
