#!/bin/bash
# Round 6, item 1: per-launch reproducibility of the condition encoder's GEMM shapes, default library and A/B knobs.
set -u
cd "$(dirname "$0")/.."
run() { echo "=== $*"; env "$@" timeout 900 python tools/r06_gemm_determinism.py 2>&1 | grep -v "amdgpu.ids"; }
run A=0
run ACE355_GEMM_BIG=2
run ACE355_GEMM_DEEP=0 ACE355_GEMM_MT1=0
