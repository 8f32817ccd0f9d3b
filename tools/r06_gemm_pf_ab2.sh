#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_gemm_pf_ab2.txt
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
line "b1 off" ACE355_GEMM_PF=0 -- --batch 1 --no-vae
for X in 16 24 32 48 64; do for CAP in 4096 2048 1024; do
  line "b1 pfx=$X cap=${CAP}KB" ACE355_GEMM_PFX=$X ACE355_GEMM_PF_CAP_KB=$CAP -- --batch 1 --no-vae
done; done
line "b1 off" ACE355_GEMM_PF=0 -- --batch 1 --no-vae
line "b2 default chains, off" ACE355_GEMM_PF=0 -- --batch 2
line "b2 default chains, on (blocked under two chains)" ACE355_GEMM_PFX=24 -- --batch 2
line "b2 one chain, off" ACE355_GEMM_PF=0 ACE355_DUAL=0 -- --batch 2
line "b2 one chain, on pfx=24" ACE355_GEMM_PFX=24 ACE355_DUAL=0 -- --batch 2
line "b2 one chain, on pfx=48" ACE355_GEMM_PFX=48 ACE355_DUAL=0 -- --batch 2
