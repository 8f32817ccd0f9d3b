#!/bin/bash
# Cache policy of the residual epilogue's single-use traffic (gemm.hip: ACE355_EPI_NT bit mask), same-box ABAB of variant builds
# against the product library.  Variants: csrc/_variants/libace355_nt{1,2,4,3,7}.so (built with -DACE355_EPI_NT=n).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
V="ace-step-1.5-for-windows_amd/csrc/_variants"
{
for nt in ${NTS:-7 1 2 4}; do
  echo "== ACE355_EPI_NT=$nt (A = product, B = variant)"
  bash tools/ab_lib.sh $V/libace355_nt$nt.so ${ROUNDS:-2} -- --steps 8 --warmup 2
done
} 2>&1 | tee $OUT/r05_epi_nt_ab.txt
