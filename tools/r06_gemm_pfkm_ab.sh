#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_gemm_pfkm_ab.txt
cd $ROOT
line() {
  local label=$1; shift
  local envs=()
  while [[ $1 != -- ]]; do envs+=("$1"); shift; done
  shift
  local ms=$(env "${envs[@]}" python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
  echo "$label: $ms ms" | tee -a $OUT
}
echo "# lib_src_sha $(python -c 'import bench; print(bench.library_source_sha())')" > $OUT
for rep in 1 2; do
line "b1 product (row-major, head-norm + residual)" ACE355_GEMM_PF=1 -- --batch 1 --no-vae
line "b1 K-major for residual launches (mode 2)" ACE355_GEMM_PF_KMAJOR=4 -- --batch 1 --no-vae
line "b1 K-major for residual + SwiGLU" ACE355_GEMM_PF_KMAJOR=12 -- --batch 1 --no-vae
line "b1 K-major for residual + SwiGLU + head-norm" ACE355_GEMM_PF_KMAJOR=28 -- --batch 1 --no-vae
line "b1 K-major residual + SwiGLU, 1 MB" ACE355_GEMM_PF_KMAJOR=12 ACE355_GEMM_PF_KM_KB=1024 -- --batch 1 --no-vae
line "b1 K-major residual + SwiGLU, 3 MB" ACE355_GEMM_PF_KMAJOR=12 ACE355_GEMM_PF_KM_KB=3072 -- --batch 1 --no-vae
line "b1 K-major SwiGLU only" ACE355_GEMM_PF_KMAJOR=8 -- --batch 1 --no-vae
done
line "b1 no prefetch" ACE355_GEMM_PF=0 -- --batch 1 --no-vae
