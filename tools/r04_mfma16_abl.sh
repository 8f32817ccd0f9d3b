#!/bin/bash
# Round 4: timing / power ablation - the GEMM K loops on 16x16x32 MFMAs (WRONG numerics: two quarter-accumulators per 32x32 tile), same
# pipe time and operand traffic, half the accumulator register traffic per FLOP.  A = the product library, B = the ablation build.
TAG=${1:-r04_v16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "== ABAB A = product, B = 16x16x32 ablation (DiT only, 8 songs)" | tee $OUT/${TAG}_ab.txt
bash tools/ab_lib.sh tools/_ab/lib_mfma16_abl.so 3 -- --steps 6 --warmup 2 --no-vae 2>&1 | tee -a $OUT/${TAG}_ab.txt
cp tools/_ab/lib_mfma16_abl.so /tmp/_abl.so; cp ace-step-1.5-for-windows_amd/csrc/libace355.so /tmp/_prod.so
cp /tmp/_abl.so ace-step-1.5-for-windows_amd/csrc/libace355.so
bash tools/gemm_clk_inpass.sh ${TAG}_abl > /dev/null; grep -E "^ *6000 " $OUT/${TAG}_abl_gemm_clk_inpass.txt
cp /tmp/_prod.so ace-step-1.5-for-windows_amd/csrc/libace355.so
