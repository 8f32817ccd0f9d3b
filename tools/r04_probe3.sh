#!/bin/bash
TAG=${1:-r04_v6}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python tools/dual_vae_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_dual_vae.txt
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "dual_chain or cfg_fork or bench_request or tiny_sampler or long_config_sampler" 2>&1 | grep -v amdgpu.ids | tail -30 > $OUT/${TAG}_tests.log
tail -14 $OUT/${TAG}_tests.log
