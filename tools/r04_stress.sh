#!/bin/bash
# Round 4: the whole GPU suite with the two-chain sampler FORCED for every request of >= 2 songs, then with the per-layer CFG fork forced
# (one chain): shakes the side-stream paths through every sampler test.  bash tools/r04_stress.sh TAG
TAG=${1:-r04_stress}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
ACE355_DUAL=2 timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40 > $OUT/${TAG}_dual2.log
tail -6 $OUT/${TAG}_dual2.log
ACE355_DUAL=0 ACE355_CFG_FORK=2 timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -40 > $OUT/${TAG}_fork2.log
tail -6 $OUT/${TAG}_fork2.log
