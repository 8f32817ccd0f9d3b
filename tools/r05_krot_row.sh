#!/bin/bash
# Row stagger of the K rotation (ACE355_GEMM_KROT_ROW = K steps between the region rows of an XCD): in-pass probes per setting, then ABAB.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for d in 0 1 2 3; do
  echo "== ACE355_GEMM_KROT_ROW=$d"
  ACE355_GEMM_KROT_ROW=$d bash tools/gemm_clk_inpass.sh r05_krotrow$d | grep -E "^ *(M|6000 +2048) "
done 2>&1 | tee $OUT/r05_krot_row_probe.txt
bash tools/ab_env.sh ACE355_GEMM_KROT_ROW=0 ACE355_GEMM_KROT_ROW=${D1:-1} 3 -- --steps 8 --warmup 2 2>&1 | tee $OUT/r05_krot_row_ab.txt
bash tools/ab_env.sh ACE355_GEMM_KROT_ROW=0 ACE355_GEMM_KROT_ROW=${D2:-2} 3 -- --steps 8 --warmup 2 2>&1 | tee -a $OUT/r05_krot_row_ab.txt
