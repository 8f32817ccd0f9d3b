#!/bin/bash
TAG=${1:-r04_v9}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== batch 2 with decode: one chain / two chains, ABAB x3" | tee $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_DUAL=0 ACE355_DUAL=1 3 -- --steps 10 --warmup 3 --batch 2 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== batch 4 with decode: one chain / two chains, ABAB x3" | tee -a $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_DUAL=0 ACE355_DUAL=1 3 -- --steps 8 --warmup 2 --batch 4 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== batch 3 with decode (2 + 1)" | tee -a $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_DUAL=0 ACE355_DUAL=1 2 -- --steps 8 --warmup 2 --batch 3 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== batch 8: rasterisation group height 4 (default) / 8, ABAB x2" | tee -a $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_GEMM_GROUPM=4 ACE355_GEMM_GROUPM=8 2 -- --steps 6 --warmup 2 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== batch 8: group height 2" | tee -a $OUT/${TAG}_ab.txt
bash tools/ab_env.sh ACE355_GEMM_GROUPM=4 ACE355_GEMM_GROUPM=2 1 -- --steps 6 --warmup 2 2>&1 | tee -a $OUT/${TAG}_ab.txt
