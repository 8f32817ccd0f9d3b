#!/bin/bash
# In-pass clock probes on a diagnostic build of the library: tools/r04_abl_lib.sh <lib.so> <tag>
cd "$(dirname "$0")/.."
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_prod.so; cp $1 $LIB
bash tools/gemm_clk_inpass.sh $2 > /dev/null 2>&1
cp /tmp/_prod.so $LIB
grep -E "^ *(M|3000|6000) " gpurun_out/${2}_gemm_clk_inpass.txt
