#!/bin/bash
# Round 6, item 1: reproducibility of the 4-wave two-stage kernels under the four MFMA orders (variants built by tools/build_variant.sh moN gemm.hip -DACE355_MFMA_ORDER=N)
set -u
cd "$(dirname "$0")/.."
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_prod.so
export ACE355_GEMM_BIG=0 SHAPES=wide REPS=30
for v in prod mo0 mo1 mo3; do
  if [ $v = prod ]; then cp /tmp/_prod.so $LIB; else cp tools/_ab/libace355_$v.so $LIB; fi
  echo "== $v"
  python tools/r06_gemm_determinism.py 2>&1 | grep -v amdgpu.ids | sed -n '1,4p;13p'
done
cp /tmp/_prod.so $LIB
