#!/bin/bash
# Round 6, small requests: next-layer weight prefetch on the side stream (ace355_dit::pf, ACE355_PREFETCH) on / off, same box, interleaved.
# Usage: gpurun -- "bash tools/r06_prefetch_ab.sh"  -> gpurun_out/r06_prefetch_ab.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_prefetch_ab.txt
mkdir -p $ROOT/gpurun_out
cd $ROOT
line() {  # label, env..., -- bench args
  local label=$1; shift
  local envs=()
  while [[ $1 != -- ]]; do envs+=("$1"); shift; done
  shift
  local ms=$(env "${envs[@]}" python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))")
  echo "$label: $ms ms" | tee -a $OUT
}
echo "# lib_src_sha $(python -c 'import bench; print(bench.library_source_sha())')" > $OUT
for rep in 1 2 3; do
  line "one song, DiT only, prefetch off" ACE355_PREFETCH=0 -- --batch 1 --no-vae
  line "one song, DiT only, prefetch on (64 wgs)" ACE355_PREFETCH=1 -- --batch 1 --no-vae
done
line "one song, DiT only, prefetch on, 32 wgs" ACE355_PREFETCH_WGS=32 -- --batch 1 --no-vae
line "one song, DiT only, prefetch on, 128 wgs" ACE355_PREFETCH_WGS=128 -- --batch 1 --no-vae
line "one song, DiT only, prefetch on, 256 wgs" ACE355_PREFETCH_WGS=256 -- --batch 1 --no-vae
for rep in 1 2; do
  line "one song with decode, prefetch off" ACE355_PREFETCH=0 -- --batch 1
  line "one song with decode, prefetch on" ACE355_PREFETCH=1 -- --batch 1
  line "two songs with decode, default (two chains, no prefetch)" ACE355_PREFETCH=0 -- --batch 2
  line "two songs with decode, one chain, no prefetch" ACE355_PREFETCH=0 ACE355_DUAL=0 -- --batch 2
  line "two songs with decode, one chain + prefetch" ACE355_PREFETCH=1 ACE355_DUAL=0 -- --batch 2
done
line "configs[0] (10 s, 10 steps), prefetch off" ACE355_PREFETCH=0 -- --batch 1 --no-vae --duration 10 --infer-steps 10
line "configs[0] (10 s, 10 steps), prefetch on" ACE355_PREFETCH=1 -- --batch 1 --no-vae --duration 10 --infer-steps 10
line "configs[0] (10 s, 10 steps), prefetch off" ACE355_PREFETCH=0 -- --batch 1 --no-vae --duration 10 --infer-steps 10
line "configs[0] (10 s, 10 steps), prefetch on" ACE355_PREFETCH=1 -- --batch 1 --no-vae --duration 10 --infer-steps 10
