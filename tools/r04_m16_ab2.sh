#!/bin/bash
# 16x16x32 port, same-box ABAB against the 32x32x16 build at three request sizes + in-pass clocks
cd "$(dirname "$0")/.."
O=gpurun_out/${AB_TAG:-r04_m16b}
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -3 > ${O}_ktests.log
echo "== batch 8 with decode (metric)" > ${O}_ab.txt
bash tools/ab_lib.sh ${AB_BASE:-tools/_ab/lib_r04_pre_hoist.so} 3 -- --steps 6 --warmup 2 >> ${O}_ab.txt 2>&1
echo "== batch 1, DiT only" >> ${O}_ab.txt
bash tools/ab_lib.sh ${AB_BASE:-tools/_ab/lib_r04_pre_hoist.so} 3 -- --steps 10 --warmup 3 --batch 1 --no-vae >> ${O}_ab.txt 2>&1
echo "== batch 2 with decode" >> ${O}_ab.txt
bash tools/ab_lib.sh ${AB_BASE:-tools/_ab/lib_r04_pre_hoist.so} 2 -- --steps 8 --warmup 2 --batch 2 >> ${O}_ab.txt 2>&1
bash tools/gemm_clk_inpass.sh ${AB_TAG:-r04_m16b} > /dev/null 2>&1
cat ${O}_ktests.log ${O}_ab.txt
