#!/bin/bash
# The other BASELINE.json configs on the final code of round 5 (one box, one command each): lines kept as profiles/r05/r05_final_cfg*_line.json
cd "$(dirname "$0")/.."
O=gpurun_out
run() { n=$1; shift; python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 > $O/r05_final_${n}_line.json; python -c "import json; d=json.load(open('$O/r05_final_${n}_line.json')); print('$n', round(d['ms_per_step'],1), 'ms', round(d['value'],3), d['unit'])"; }
run cfg0 --steps 20 --warmup 5 --batch 1 --no-vae --duration 10 --infer-steps 10
run cfg1 --steps 10 --warmup 3 --batch 1 --no-vae
run cfg2 --steps 4 --warmup 1 --duration 120
run cfg2_fp8 --steps 4 --warmup 1 --duration 120 --fp8
run cfg3_share --steps 2 --warmup 1 --duration 240 --infer-steps 60 --batch 4
run cfg4_share_bf16 --steps 2 --warmup 1 --duration 600 --batch 8
run cfg4_share_fp8 --steps 2 --warmup 1 --duration 600 --batch 8 --fp8
run metric_fp8 --steps 8 --warmup 3 --fp8
