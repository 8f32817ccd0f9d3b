#!/bin/bash
# A = product, B = each diagnostic variant library in turn (tools/_ab/lib_<name>.so), 3 pairs each, quiet-host bench
cd "$(dirname "$0")/.."
for n in "$@"; do echo "== $n"; bash tools/ab_lib.sh tools/_ab/lib_$n.so 3 -- --steps 8 --warmup 3; done
