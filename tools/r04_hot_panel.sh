#!/bin/bash
# What does the fabric cost a K step?  In-pass clock probes with the A / W panels of every workgroup replaced by tile 0's (L2-resident):
# ACE355_GEMM_CLK = 1 product, 3 A hot, 5 W hot, 7 both hot (results are wrong by construction; only the probes are read)
cd "$(dirname "$0")/.."
for v in 1 3 5 7 1; do
  ACE355_CLK_VAL=$v bash tools/gemm_clk_inpass.sh r04_hot_$v > /dev/null 2>&1
  echo "== ACE355_GEMM_CLK=$v"; grep -E "^ *(3000|6000) " gpurun_out/r04_hot_${v}_gemm_clk_inpass.txt
done > gpurun_out/r04_hot_panel.txt 2>&1
cat gpurun_out/r04_hot_panel.txt
