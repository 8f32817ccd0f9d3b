#!/usr/bin/env python
"""Per-kernel breakdown of a steady-state window of a rocprofv3 kernel trace: launches, average duration, share of the span, and
the idle time between consecutive kernels.  Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py ...;
python tools/trace_breakdown.py DIR [first_fraction last_fraction]"""
import csv, glob, re, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * lo):int(len(rows) * hi)]
sh = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("ace355::(anonymous namespace)::", "").replace("ace355::", ""))[:64]
agg = defaultdict(lambda: [0, 0])
gap = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 0:
        gap += g
for r in rows:
    k = f'{sh(r["Kernel_Name"])} grid={r.get("Grid_Size_X", "?")}x{r.get("Grid_Size_Y", "?")} wg={r.get("Workgroup_Size_X", "?")}'
    agg[k][0] += 1
    agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(v[1] for v in agg.values())
print(f"window: {len(rows)} launches, span {span/1e6:.2f} ms, busy {busy/1e6:.2f} ms, idle between kernels {gap/1e6:.2f} ms ({100*gap/span:.1f} %)")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{100*t/span:5.1f} %  {n:6d} x {t/n/1e3:7.2f} us  {k}")
