#!/bin/bash
# Same-box ABAB of the MFMA instruction order inside a half K step (gemm.hip: ACE355_MFMA_ORDER, compile-time; the sums are the same sums).
# Variants built in the container: tools/build_variant.sh mo1 gemm.hip -DACE355_MFMA_ORDER=1; ... mo2 ... =2
cd "$(dirname "$0")/.."
V=ace-step-1.5-for-windows_amd/csrc/_variants
for n in ${VARIANTS:-mo1 mo2}; do
  echo "== A = order 0 (product), B = $n"
  bash tools/ab_lib.sh $V/libace355_$n.so 3 -- --steps 10 --warmup 3
done
