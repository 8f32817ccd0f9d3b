#!/bin/bash
cd "$(dirname "$0")/.."
ACE355_BENCH_TRACE=1 python bench.py --no-cpu-baseline --no-roofline --steps 40 --warmup 3 2>&1 | grep -E "bench trace" | awk '{print}' > gpurun_out/r04_pass_trace.txt
python - <<'PY'
import re
rows=[l for l in open('gpurun_out/r04_pass_trace.txt')]
tot=[float(re.search(r'pass gpu ([\d.]+)',l).group(1)) for l in rows]
med=sorted(tot)[len(tot)//2]
print('passes',len(tot),'median',med)
for l,t in zip(rows,tot):
    if t>med*1.04: print(l.strip())
print('--- a normal pass:'); print(rows[1].strip())
PY
