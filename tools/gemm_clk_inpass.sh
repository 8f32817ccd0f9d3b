#!/bin/bash
# In-pass clock probes of every GEMM launch of one request (ACE355_GEMM_CLK=1): per (M, N, K, mode) the mean shader clock, cycles per K
# step, prologue and epilogue cycles.  bash tools/gemm_clk_inpass.sh TAG [bench args]
TAG=${1:-r04_clk}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
ACE355_SAMPLE_GRAPH=0 ACE355_GEMM_CLK=${ACE355_CLK_VAL:-1} python bench.py --steps 1 --warmup 1 --no-vae --no-roofline --no-cpu-baseline "$@" > /dev/null 2> /tmp/clk_raw.txt
python - > $OUT/${TAG}_gemm_clk_inpass.txt <<'PY'
import re, collections
rows = collections.defaultdict(list)
pat = re.compile(r"M=(\d+) N=(\d+) K=(\d+) mode=(\d+) kparts=(\d+) tiles=(\d+): ([\d.]+) GHz shader clock, (\d+) cycles / K-step \(([\d.]+) us\) x (\d+); prologue (\d+), epilogue issue (\d+) / acked (\d+)")
for line in open("/tmp/clk_raw.txt"):
    m = pat.search(line)
    if m:
        g = m.groups()
        rows[tuple(int(x) for x in g[:6])].append([float(g[6]), float(g[7]), float(g[8]), int(g[9]), float(g[10]), float(g[11]), float(g[12])])
print("in-pass GEMM clock probes (workgroup 0 of every launch of the second pass included; means over the launches of one shape)")
print(f"{'M':>5} {'N':>6} {'K':>5} mode kparts tiles  launches   GHz  cyc/Kstep  us/Kstep  Ksteps  prologue  epi-issue  epi-acked  epi-acked-us")
for k, v in sorted(rows.items()):
    n = len(v)
    mean = [sum(x[i] for x in v) / n for i in range(7)]
    print(f"{k[0]:5d} {k[1]:6d} {k[2]:5d} {k[3]:4d} {k[4]:6d} {k[5]:5d}  {n:8d}  {mean[0]:5.3f}  {mean[1]:9.0f}  {mean[2]:8.3f}  {mean[3]:6.0f}  {mean[4]:8.0f}  {mean[5]:9.0f}  {mean[6]:9.0f}  {mean[6] / (mean[0] * 1e3):10.2f}")
PY
cat $OUT/${TAG}_gemm_clk_inpass.txt
