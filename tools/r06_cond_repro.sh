#!/bin/bash
# Round 6, verdict item 1: the full-size condition encoder under the library's knobs, one fresh process per variant.
# Usage (GPU box): bash tools/r06_cond_repro.sh > gpurun_out/r06_cond_repro.txt 2>&1
set -u
cd "$(dirname "$0")/.."
O=/tmp/r06_cond_oracle.pt
run() {
    echo "=== $*"
    env "$@" timeout 600 python tools/r06_cond_repro.py --reps 4 --oracle $O 2>&1 | grep -v "^$"
}
run A=0
run A=0
run ACE355_GEMM_KSPLIT=1
run ACE355_GEMM_MT1=0
run ACE355_GEMM_KROT=0
run ACE355_GEMM_DEEP=0
run ACE355_GEMM=v1
run ACE355_GEMM_HEADEPI=0
run ACE355_ATTN_GQA=0
echo "=== dirty memory"
timeout 600 python tools/r06_cond_repro.py --reps 3 --oracle $O --dirty 8 2>&1 | grep -v "^$"
