#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_gemm_pfself_ab.txt
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
line "b1: no prefetch at all" ACE355_GEMM_PF=0 -- --batch 1 --no-vae
line "b1: next-launch prefetch (product)" ACE355_GEMM_PF=1 -- --batch 1 --no-vae
for X in 2 4 8 16; do
  line "b1: self prefetch $X per XCD, every projection, no next-launch prefetch" ACE355_GEMM_PF=0 ACE355_GEMM_PFSELF=$X -- --batch 1 --no-vae
  line "b1: self prefetch $X per XCD + next-launch prefetch" ACE355_GEMM_PF=1 ACE355_GEMM_PFSELF=$X -- --batch 1 --no-vae
done
for X in 4 8; do
  line "b1: self prefetch $X per XCD for SwiGLU only + next-launch" ACE355_GEMM_PFSELF=$X ACE355_GEMM_PFSELF_MODES=8 -- --batch 1 --no-vae
  line "b1: self prefetch $X per XCD for SwiGLU + residual + next-launch" ACE355_GEMM_PFSELF=$X ACE355_GEMM_PFSELF_MODES=12 -- --batch 1 --no-vae
done
line "b1: no prefetch at all" ACE355_GEMM_PF=0 -- --batch 1 --no-vae
line "b1: next-launch prefetch (product)" ACE355_GEMM_PF=1 -- --batch 1 --no-vae
