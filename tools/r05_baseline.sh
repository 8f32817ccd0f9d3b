#!/bin/bash
# Round-5 session baseline on one box: GPU suite, the default bench line, kernel stats of the bench command, in-pass GEMM clock probes.
TAG=${1:-r05_base}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5 > $OUT/${TAG}_gpu_pytest.log
cat $OUT/${TAG}_gpu_pytest.log
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line.json
python -c "
import json; d=json.load(open('$OUT/${TAG}_bench_line.json')); r=d['roofline']
print('bench', round(d['value'],3), 'songs/s', round(d['ms_per_step'],1), 'ms; gemm frac', round(r['frac'],4), 'gemm ms', round(r['gemm_ms_per_pass'],1), 'attn ms', round(r['attn_ms_per_pass'],1), 'vae ms', round(r['vae_conv_ms_per_pass'],1))"
bash tools/ab_env.sh ACE355_GEMM_BIG=1 ACE355_GEMM_BIG=0 1 -- --steps 8 --warmup 2
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_stats.log 2>&1
cp "$(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1)" $OUT/${TAG}_bench_kernel_stats.csv
python $ROOT/tools/trace_breakdown.py /tmp/prof_stats 0.45 0.9 > $OUT/${TAG}_trace_breakdown.txt 2>&1
head -45 $OUT/${TAG}_trace_breakdown.txt
cd $ROOT
bash tools/gemm_clk_inpass.sh $TAG
