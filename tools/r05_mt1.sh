#!/bin/bash
# 64 x 128 tiles for launches whose 128-row form leaves more than half the chip idle (ACE355_GEMM_MT1): tests + ABAB at 1 / 2 songs and configs[0]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
{
bash tools/ab_env.sh ACE355_GEMM_MT1=0 ACE355_GEMM_MT1=1 3 -- --steps 10 --warmup 3 --batch 1 --no-vae
bash tools/ab_env.sh ACE355_GEMM_MT1=0 ACE355_GEMM_MT1=1 2 -- --steps 10 --warmup 3 --batch 2 --no-vae
bash tools/ab_env.sh ACE355_GEMM_MT1=0 ACE355_GEMM_MT1=1 2 -- --steps 20 --warmup 5 --batch 1 --no-vae --duration 10 --infer-steps 10
} 2>&1 | tee $OUT/r05_mt1_ab.txt
