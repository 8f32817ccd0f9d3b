#!/bin/bash
# Round 6, small requests: the next projection's weight rows prefetched into the right XCD's L2 by spare workgroups of the current GEMM launch
# (GemmEpilogue::pf_*, ACE355_GEMM_PF / ACE355_GEMM_PFX), on / off, same box, interleaved.  -> gpurun_out/r06_gemm_pf_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_gemm_pf_ab.txt
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
for rep in 1 2 3; do
  line "one song, DiT only, prefetch off" ACE355_GEMM_PF=0 -- --batch 1 --no-vae
  line "one song, DiT only, prefetch on (8 per XCD)" ACE355_GEMM_PF=1 -- --batch 1 --no-vae
done
for X in 2 4 16 24; do
  line "one song, DiT only, prefetch on, $X per XCD" ACE355_GEMM_PFX=$X -- --batch 1 --no-vae
done
for rep in 1 2; do
  line "two songs with decode, prefetch off" ACE355_GEMM_PF=0 -- --batch 2
  line "two songs with decode, prefetch on" ACE355_GEMM_PF=1 -- --batch 2
  line "configs[0], prefetch off" ACE355_GEMM_PF=0 -- --batch 1 --no-vae --duration 10 --infer-steps 10
  line "configs[0], prefetch on" ACE355_GEMM_PF=1 -- --batch 1 --no-vae --duration 10 --infer-steps 10
done
line "four songs with decode, prefetch off" ACE355_GEMM_PF=0 -- --batch 4
line "four songs with decode, prefetch on (max rows 3200)" ACE355_GEMM_PF=1 ACE355_GEMM_PF_MAX_ROWS=3200 -- --batch 4
line "eight songs, prefetch off" ACE355_GEMM_PF=0 --
line "eight songs, prefetch on (default: nothing qualifies)" ACE355_GEMM_PF=1 --
