#!/bin/bash
# Round-end evidence in one gpurun call: kernel-trace stats and the two PMC passes (each in its own run, as the MI355X guide
# prescribes) of the headline bench command, summaries copied to gpurun_out/ for profiles/.  Usage: bash tools/profile_round.sh TAG
TAG=${1:-r01_vX}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- $CMD > $OUT/${TAG}_stats.log 2>&1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
tail -1 $OUT/${TAG}_stats.log | cut -c1-400
for C in FETCH_SIZE WRITE_SIZE; do
  # counters restricted to the GEMM / conv / norm families: with every kernel instrumented rocprofv3 7.2 segfaults inside the
  # tool at the first attn3_kernel dispatch (reproducible on this image; the kernel-trace pass above is unaffected)
  timeout 900 rocprofv3 --pmc $C --kernel-include-regex "gemm_sp_kernel|conv_kernel|rmsnorm" --output-format csv -d /tmp/prof_$C -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_$C.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/prof_$C $OUT/${TAG}_pmc_$C.json
done
