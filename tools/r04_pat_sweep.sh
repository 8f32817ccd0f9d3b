#!/bin/bash
# DMA piece placement sweep: in-pass K-step probes + bench for each diagnostic build given (tools/_ab/lib_<name>.so)
cd "$(dirname "$0")/.."
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_prod.so
OUT=gpurun_out/r04_pat_sweep.txt
mkdir -p gpurun_out
: > $OUT
run() {  # name
  bash tools/gemm_clk_inpass.sh r04_pat_$1 > /dev/null 2>&1
  echo "== $1: in-pass probes" >> $OUT; grep -E "^ *(3000|6000) +(2048|4096|12288) +(2048|6144) " gpurun_out/r04_pat_$1_gemm_clk_inpass.txt >> $OUT
  python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 bench', round(d['ms_per_step'],2), 'ms')" >> $OUT
}
run prod
for n in "$@"; do cp tools/_ab/lib_$n.so $LIB; run $n; cp /tmp/_prod.so $LIB; done
run prod
cat $OUT
