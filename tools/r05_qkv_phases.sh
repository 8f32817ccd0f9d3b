#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
ACE355_SAMPLE_GRAPH=0 ACE355_GEMM_CLK=1 python bench.py --steps 1 --warmup 1 --no-vae --no-roofline --no-cpu-baseline > /dev/null 2> /tmp/clk_raw.txt
python - <<'PY'
import re, collections
rows = collections.defaultdict(list)
cur=None
for line in open("/tmp/clk_raw.txt"):
    m = re.search(r"M=(\d+) N=(\d+) K=(\d+) mode=(\d+).*prologue (\d+), epilogue issue (\d+) / acked (\d+)", line)
    if m: cur=(m.group(1),m.group(2),m.group(3),m.group(4)); tot=int(m.group(7)); continue
    m = re.search(r"epilogue phases \(wave 0\): sums exchanged (\d+), staged (\d+)", line)
    if m and cur: rows[cur].append((int(m.group(1)), int(m.group(2)), tot))
for k,v in sorted(rows.items()):
    n=len(v); print(k, n, [round(sum(x[i] for x in v)/n) for i in range(3)])
PY
