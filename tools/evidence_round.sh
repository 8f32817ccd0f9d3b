#!/bin/bash
# Clock / power evidence for the GEMM roofline fraction, one gpurun call:
#   1. in-kernel shader-clock probe of the four big projections (ACE355_GEMM_CLK=1, tools/gemm_clk.py)
#   2. rocm-smi sclk / power samples taken WHILE the headline bench runs, plus an idle sample before it
# Usage: bash tools/evidence_round.sh TAG   -> gpurun_out/${TAG}_gemm_clk.txt, ${TAG}_smi_trace.txt, ${TAG}_bench.json
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
ACE355_GEMM_CLK=1 python tools/gemm_clk.py 2> $OUT/${TAG}_gemm_clk_raw.txt
python - "$OUT/${TAG}_gemm_clk_raw.txt" > $OUT/${TAG}_gemm_clk.txt <<'PY'
import re, sys, collections
rows = collections.OrderedDict()
name = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("[ace355 gemm clk]"):
        rows.setdefault(name, []).append(line)
    elif line:
        name = line
for name, ls in rows.items():
    print(f"== {name}: {len(ls)} launches, random bf16 operands, M=6000; last 3 (warm):")
    for l in ls[-3:]:
        print("   " + l)
PY
echo "# idle (before the bench)" > $OUT/${TAG}_smi_trace.txt
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" >> $OUT/${TAG}_smi_trace.txt
python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err &
BP=$!
sleep 12   # model build + warm-up
echo "# under load (bench.py --steps 40, sampled every ~0.3 s)" >> $OUT/${TAG}_smi_trace.txt
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $OUT/${TAG}_smi_trace.txt
  echo >> $OUT/${TAG}_smi_trace.txt
  sleep 0.2
done
wait $BP
tail -c 600 $OUT/${TAG}_bench.json
