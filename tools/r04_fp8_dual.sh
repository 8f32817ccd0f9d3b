#!/bin/bash
TAG=${1:-r04_v11}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== MXFP8 mode, batch 8 with decode: one chain / two chains / two chains planned for 128 CUs" | tee $OUT/${TAG}_fp8_dual.txt
for r in 1 2; do
for V in "ACE355_DUAL=0" "ACE355_DUAL=2" "ACE355_DUAL=2 ACE355_DUAL_SLOTS_MIN_ROWS=1536"; do
  env $V python bench.py --fp8 --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_fp8_dual.txt
done; done
python -m pytest tests -m gpu -x -q -s -k "cover_switch or dual_chain or cfg_fork" 2>&1 | grep -v amdgpu.ids | tail -8
