#!/bin/bash
# Row sums of the folded-norm consumers requested under the last K step (gemm.hip: ACE355_EPI_PRESQ): in-pass clock probes of both builds,
# the GEMM kernel tests on the product build, then a same-box ABAB (A = product = PRESQ 1, B = csrc/_variants/libace355_presq0.so).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
V="ace-step-1.5-for-windows_amd/csrc/_variants/libace355_${VAR:-presq0}.so"
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/gemm_clk_inpass.sh r05_${VAR:-presq0}_A | grep -E "^ *(M|3000|6000) "
bash tools/r04_abl_lib.sh $V r05_${VAR:-presq0}_B
bash tools/ab_lib.sh $V ${ROUNDS:-3} -- --steps 8 --warmup 2 2>&1 | tee $OUT/r05_${VAR:-presq0}_ab.txt
