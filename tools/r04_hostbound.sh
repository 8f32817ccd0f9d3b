#!/bin/bash
TAG=${1:-r04_v15}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
nproc; lscpu | grep -E "Model name|MHz" | head -3
echo "== batch 2 with decode: chains x graph replay" | tee $OUT/${TAG}_ab.txt
for r in 1 2; do
for V in "ACE355_DUAL=0 ACE355_SAMPLE_GRAPH=0" "ACE355_DUAL=1 ACE355_SAMPLE_GRAPH=0" "ACE355_DUAL=0 ACE355_SAMPLE_GRAPH=1" "ACE355_DUAL=1 ACE355_SAMPLE_GRAPH=1"; do
  env $V python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --batch 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_ab.txt
done; done
echo "== batch 3" | tee -a $OUT/${TAG}_ab.txt
for V in "ACE355_DUAL=0 ACE355_SAMPLE_GRAPH=0" "ACE355_DUAL=1 ACE355_SAMPLE_GRAPH=0" "ACE355_DUAL=1 ACE355_SAMPLE_GRAPH=1"; do
  env $V python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 3 --batch 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_ab.txt
done
