#!/bin/bash
TAG=${1:-r04_v14}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
ACE355_GEMM_KROT=2 timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q 2>&1 | grep -v amdgpu.ids | tail -4 | tee $OUT/${TAG}_tests.log
echo "== ABAB K rotation off / N<=2048 launches / all launches" | tee $OUT/${TAG}_ab.txt
for r in 1 2 3; do for V in 0 1 2; do
  ACE355_GEMM_KROT=$V python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('KROT=$V', round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_ab.txt
done; done
ACE355_GEMM_KROT=1 bash tools/gemm_clk_inpass.sh ${TAG}_krot1 > /dev/null; grep -E "^ *(3000|6000) " $OUT/${TAG}_krot1_gemm_clk_inpass.txt
