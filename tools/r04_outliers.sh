#!/bin/bash
# Where do the occasional 10-25 % slow bench invocations come from?  Ten default-size invocations on one box, per-pass spread of each.
cd "$(dirname "$0")/.."
for i in $(seq 10); do
  python bench.py --no-cpu-baseline --no-roofline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['step_ms_spread']; g=d.get('gpu_state_under_load',{})
print('run $i', round(d['ms_per_step'],1), 'ms; passes min/med/max', round(s['min'],1), round(s['median'],1), round(s['max'],1), '|', {k:g[k] for k in list(g)[:6]})"
done
