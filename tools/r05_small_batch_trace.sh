#!/bin/bash
# Per-kernel breakdown (kernel trace) of requests of 2 / 4 songs, one chain (ACE355_DUAL=0) - which tile each launch takes and what it costs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 2 4; do
  rm -rf /tmp/prof_b$b
  ACE355_DUAL=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b$b -- python $ROOT/bench.py --steps 3 --warmup 1 --batch $b --no-vae --no-cpu-baseline --no-roofline > /dev/null 2>&1
  echo "== batch $b, one chain"
  python $ROOT/tools/trace_breakdown.py /tmp/prof_b$b 0.3 0.9 | head -13
done 2>&1 | tee $OUT/r05_small_batch_kernel_breakdown.txt
