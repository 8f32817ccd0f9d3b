#!/bin/bash
# Round 4: the dual-chain sampler in the product path.  Parity of the new contract, then same-box ABAB of bench.py with one chain /
# two chains (and two chains planned for the whole chip), at batch 8 / 4 / 2.  bash tools/r04_dual_bench.sh TAG
TAG=${1:-r04_v5}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "dual_chain or cfg_fork or bench_request or tiny_sampler or long_config_sampler" 2>&1 | grep -v amdgpu.ids | tail -30 > $OUT/${TAG}_tests.log
tail -14 $OUT/${TAG}_tests.log
echo "== batch 8 with decode: ABAB one chain / two chains" | tee $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_DUAL=0 ACE355_DUAL=1 2 -- --steps 6 --warmup 2 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== two chains, launches planned for the whole chip" | tee -a $OUT/${TAG}_ab.txt
ACE355_DUAL_SLOTS_MIN_ROWS=1000000 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_ab.txt
for B in 4 2; do
  echo "== batch $B with decode" | tee -a $OUT/${TAG}_ab.txt
  bash tools/ab_env.sh ACE355_DUAL=0 ACE355_DUAL=1 1 -- --steps 8 --warmup 2 --batch $B 2>&1 | tee -a $OUT/${TAG}_ab.txt
  ACE355_DUAL_SLOTS_MIN_ROWS=64 python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 --batch $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('slots for every chain:', round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_ab.txt
done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line.json
python -c "
import json; d=json.load(open('$OUT/${TAG}_bench_line.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], {k: r[k] for k in ('achieved','frac','avg_launch_us','gemm_ms_per_pass','concurrent_chains','attn_tflops','attn_ms_per_pass','vae_conv_ms_per_pass')})"
