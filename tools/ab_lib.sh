#!/bin/bash
# Same-box ABAB of two builds of the library: tools/ab_lib.sh <other.so> [rounds] -- <bench.py args>
# (A = csrc/libace355.so as built, B = the other file; the box's clocks differ by 1-2 % from call to call, so only
#  alternating runs on ONE box compare two builds)
set -e
cd "$(dirname "$0")/.."
OTHER=$1; shift
ROUNDS=2
if [ "$1" != "--" ]; then ROUNDS=$1; shift; fi
shift
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_ab_A.so; cp $OTHER /tmp/_ab_B.so
for r in $(seq $ROUNDS); do
  for v in A B; do
    cp /tmp/_ab_$v.so $LIB
    python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'ms', round(d['value'],3))"
  done
done
cp /tmp/_ab_A.so $LIB
