#!/bin/bash
# tools/r04_lib_check.sh <name>: kernel tests ON the variant library tools/_ab/lib_<name>.so, then ABAB product vs variant, then the variant's K-step probes
cd "$(dirname "$0")/.."
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_prod.so; cp tools/_ab/lib_$1.so $LIB
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -2
bash tools/gemm_clk_inpass.sh r04_var_$1 > /dev/null 2>&1
cp /tmp/_prod.so $LIB
grep -E "^ *(M|3000|6000) " gpurun_out/r04_var_$1_gemm_clk_inpass.txt
bash tools/ab_lib.sh tools/_ab/lib_$1.so ${2:-4} -- --steps 8 --warmup 3
