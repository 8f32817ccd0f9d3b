#!/bin/bash
# Zero row for the GEMM tiles' pad rows (ACE355_GEMM_ZROW): tests, then same-box ABAB of the switch.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_metric_shapes_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_env.sh ACE355_GEMM_ZROW=0 ACE355_GEMM_ZROW=1 ${ROUNDS:-4} -- --steps 8 --warmup 2 2>&1 | tee $OUT/r05_zrow_ab.txt
