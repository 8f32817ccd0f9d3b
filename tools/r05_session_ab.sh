#!/bin/bash
# Same-box ABAB of the GEMM K-loop work of the last session of round 5: A = the product library, B = the same sources built with the MFMA order of rounds
# 1-5 and the half-by-half K loop (-DACE355_MFMA_ORDER=0 -DACE355_MFMA_PAIR=0: tools/build_variant.sh sessionstart gemm.hip ...) run with ACE355_GEMM_KROT_ROW=0
cd "$(dirname "$0")/.."
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_ab_A.so; cp ace-step-1.5-for-windows_amd/csrc/_variants/libace355_sessionstart.so /tmp/_ab_B.so
for B in 8 4; do
  echo "== $B songs"
  for r in 1 2 3; do
    cp /tmp/_ab_A.so $LIB
    python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --batch $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('A (serpentine + pair loops + row stagger)', round(d['ms_per_step'],2), 'ms')"
    cp /tmp/_ab_B.so $LIB
    ACE355_GEMM_KROT_ROW=0 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --batch $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B (the K loop of rounds 1-5)            ', round(d['ms_per_step'],2), 'ms')"
  done
done
cp /tmp/_ab_A.so $LIB
