#!/bin/bash
# (1) tests; (2) layer-0 CFG dedup on / off; (3) MFMA-free K loop for waves whose rows all lie beyond M (A = product, B = -DACE355_GEMM_DEADWAVE=0);
# (4) warm-up of the next launch's weights
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_metric_shapes_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -5
bash tools/ab_env.sh ACE355_DEDUP0=0 ACE355_DEDUP0=1 4 -- --steps 8 --warmup 2 2>&1 | tee $OUT/r05_dedup0_ab.txt
bash tools/ab_lib.sh ace-step-1.5-for-windows_amd/csrc/_variants/libace355_dead0.so 4 -- --steps 8 --warmup 2 2>&1 | tee $OUT/r05_deadwave_ab.txt
for w in 0 2; do
  echo "== ACE355_GEMM_WARM=$w"
  ACE355_GEMM_WARM=$w bash tools/gemm_clk_inpass.sh r05_warm$w | grep -E "^ *(M|3000|6000) "
done 2>&1 | tee $OUT/r05_warm_probe.txt
bash tools/ab_env.sh ACE355_GEMM_WARM=0 ACE355_GEMM_WARM=2 3 -- --steps 8 --warmup 2 2>&1 | tee $OUT/r05_warm_ab.txt
