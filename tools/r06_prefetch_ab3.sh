#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06_prefetch_ab3.txt
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
line "off" ACE355_PREFETCH=0 -- --batch 1 --no-vae
for SP in 1 0; do for BS in 64; do for U in 2 4 8; do for W in 128 256 512 1024; do
  line "on split=$SP bs=$BS unroll=$U wgs=$W" ACE355_PREFETCH=1 ACE355_PREFETCH_SPLIT=$SP ACE355_PREFETCH_BS=$BS ACE355_PREFETCH_UNROLL=$U ACE355_PREFETCH_WGS=$W -- --batch 1 --no-vae
done; done; done; done
line "off" ACE355_PREFETCH=0 -- --batch 1 --no-vae
