#!/bin/bash
# Same-box ABAB of an environment switch: tools/ab_env.sh VAR=a VAR=b [rounds] -- <bench.py args>
set -e
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
ROUNDS=2
if [ "$1" != "--" ]; then ROUNDS=$1; shift; fi
shift
for r in $(seq $ROUNDS); do
  for v in "$A" "$B"; do
    env $v python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), 'ms', round(d['value'],3))"
  done
done
