#!/bin/bash
# Per-kernel breakdown of a one-song request with / without the 64-row tiles (kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/prof_mt1_$v
  ACE355_GEMM_MT1=$v timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_mt1_$v -- python $ROOT/bench.py --steps 3 --warmup 1 --batch 1 --no-vae --no-cpu-baseline --no-roofline > /dev/null 2>&1
  echo "== ACE355_GEMM_MT1=$v"
  python $ROOT/tools/trace_breakdown.py /tmp/prof_mt1_$v 0.3 0.9 | head -14
done 2>&1 | tee $OUT/r05_mt1_batch1_kernel_breakdown.txt
