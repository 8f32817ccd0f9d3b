#!/bin/bash
TAG=${1:-r04_v10}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== batch 1, DiT only: per-layer CFG fork off / forced (mode 2), ABAB x2" | tee $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_CFG_FORK=0 ACE355_CFG_FORK=2 2 -- --steps 10 --warmup 3 --batch 1 --no-vae 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== configs[0]: fork off / forced" | tee -a $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_CFG_FORK=0 ACE355_CFG_FORK=2 2 -- --steps 20 --warmup 5 --batch 1 --no-vae --duration 10 --infer-steps 10 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== batch 2 with decode: one chain / one chain + fork / two chains" | tee -a $OUT/${TAG}_ab.txt
for r in 1 2; do
for V in "ACE355_DUAL=0 ACE355_CFG_FORK=0" "ACE355_DUAL=0 ACE355_CFG_FORK=2" "ACE355_DUAL=1 ACE355_CFG_FORK=0"; do
  env $V python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --batch 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_ab.txt
done; done
echo "== batch 4 with decode: one chain / one chain + fork" | tee -a $OUT/${TAG}_ab.txt
bash tools/ab_env.sh "ACE355_DUAL=0" "ACE355_CFG_FORK=2" 2 -- --steps 8 --warmup 2 --batch 4 2>&1 | tee -a $OUT/${TAG}_ab.txt
