#!/bin/bash
TAG=${1:-r04_v13}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_metric_shapes_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/${TAG}_tests.log
bash tools/gemm_clk_inpass.sh ${TAG} > /dev/null; grep -E " 2  +1 " $OUT/${TAG}_gemm_clk_inpass.txt
echo "== ABAB: A = this build (hoisted H loads), B = the build before it" | tee $OUT/${TAG}_ab.txt
bash tools/ab_lib.sh tools/_ab/lib_r04_pre_hoist.so 3 -- --steps 6 --warmup 2 2>&1 | tee -a $OUT/${TAG}_ab.txt
echo "== batch 4" | tee -a $OUT/${TAG}_ab.txt
bash tools/ab_lib.sh tools/_ab/lib_r04_pre_hoist.so 2 -- --steps 6 --warmup 2 --batch 4 2>&1 | tee -a $OUT/${TAG}_ab.txt
