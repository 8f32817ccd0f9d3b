#!/bin/bash
# Same-box ABAB of two builds of the library under an arbitrary command: tools/ab_lib_cmd.sh <other.so> <rounds> -- <command...>
set -e
cd "$(dirname "$0")/.."
OTHER=$1; ROUNDS=$2; shift 3
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_ab_A.so; cp $OTHER /tmp/_ab_B.so
for r in $(seq $ROUNDS); do
  for v in A B; do
    cp /tmp/_ab_$v.so $LIB
    echo "== $v"; "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/_ab_A.so $LIB
