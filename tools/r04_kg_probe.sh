#!/bin/bash
# Round 4: intra-workgroup split-K (gemm_sp_kernel KG = 2) for the small-M launches.  Kernel / sampler parity, then same-box ABAB at
# batch 1 (BASELINE configs[1]), configs[0] and batch 2.  bash tools/r04_kg_probe.sh TAG
TAG=${1:-r04_v7}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_dit_gpu.py tests/test_metric_shapes_gpu.py -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids | tail -60 > $OUT/${TAG}_tests.log
tail -25 $OUT/${TAG}_tests.log
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value'],3))"; }
echo "== batch 1, DiT only (configs[1])" | tee $OUT/${TAG}_ab.txt
for r in 1 2; do for V in 0 2 1; do
  ACE355_GEMM_KG=$V python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --batch 1 --no-vae 2>/dev/null | tail -1 | line "KG=$V" | tee -a $OUT/${TAG}_ab.txt
done; done
echo "== configs[0]: 10 s, 10 steps, batch 1, DiT only" | tee -a $OUT/${TAG}_ab.txt
for r in 1 2; do for V in 0 2; do
  ACE355_GEMM_KG=$V python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --batch 1 --no-vae --duration 10 --infer-steps 10 2>/dev/null | tail -1 | line "KG=$V" | tee -a $OUT/${TAG}_ab.txt
done; done
echo "== batch 2 with decode: dual x KG" | tee -a $OUT/${TAG}_ab.txt
for D in 0 1; do for V in 0 2; do
  ACE355_DUAL=$D ACE355_GEMM_KG=$V python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 2 --batch 2 2>/dev/null | tail -1 | line "DUAL=$D KG=$V" | tee -a $OUT/${TAG}_ab.txt
done; done
echo "== batch 4 with decode: dual x KG" | tee -a $OUT/${TAG}_ab.txt
for D in 0 1; do for V in 0 2; do
  ACE355_DUAL=$D ACE355_GEMM_KG=$V python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 --batch 4 2>/dev/null | tail -1 | line "DUAL=$D KG=$V" | tee -a $OUT/${TAG}_ab.txt
done; done
