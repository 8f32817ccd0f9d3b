#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_gemm_pf_ab3.txt
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
line "b1 off" ACE355_GEMM_PF=0 -- --batch 1 --no-vae
line "b1 on, every projection (default knobs 24 / 2 MB)" ACE355_GEMM_PF=1 -- --batch 1 --no-vae
line "b1 on, for the head-norm projections only (QKV, cross-q)" ACE355_GEMM_PF_MODES=16 -- --batch 1 --no-vae
line "b1 on, head-norm + residual projections" ACE355_GEMM_PF_MODES=20 -- --batch 1 --no-vae
line "b1 on, head-norm only, 32 per XCD" ACE355_GEMM_PF_MODES=16 ACE355_GEMM_PFX=32 -- --batch 1 --no-vae
line "b1 on, head-norm only, 16 per XCD" ACE355_GEMM_PF_MODES=16 ACE355_GEMM_PFX=16 -- --batch 1 --no-vae
line "b1 on, head-norm only, cap 4 MB" ACE355_GEMM_PF_MODES=16 ACE355_GEMM_PF_CAP_KB=4096 -- --batch 1 --no-vae
done
line "b1 with decode off" ACE355_GEMM_PF=0 -- --batch 1
line "b1 with decode on" ACE355_GEMM_PF=1 -- --batch 1
line "b2 off" ACE355_GEMM_PF=0 -- --batch 2
line "b2 on (default: nothing qualifies at two chains / 1500 rows)" ACE355_GEMM_PF=1 -- --batch 2
line "b3 off" ACE355_GEMM_PF=0 -- --batch 3
line "b3 on" ACE355_GEMM_PF=1 -- --batch 3
