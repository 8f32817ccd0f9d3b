#!/usr/bin/env python
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (steady-state window): where the GPU waits on launches
rather than on kernels.  Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py ...; python tools/gap_summary.py DIR"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one steady-state pass of bench.py: between the 2nd and 3rd post-decode peak_scale launches (warm-up | timed pass 1 | ...)
marks = [i for i, r in enumerate(rows) if "peak_scale_kernel" in r["Kernel_Name"]]
if len(marks) >= 3:
    rows = rows[marks[0] + 1:marks[1] + 1]
else:
    rows = rows[int(len(rows) * 0.6):]
busy = gap = 0
hist = defaultdict(lambda: [0, 0.0])
big = []
for a, b in zip(rows, rows[1:]):
    busy += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 0:
        gap += g
        k = "<1us" if g < 1000 else "<2us" if g < 2000 else "<4us" if g < 4000 else "<10us" if g < 10000 else "<100us" if g < 100000 else ">=100us"
        hist[k][0] += 1
        hist[k][1] += g
        if g >= 10000:
            big.append((g, a["Kernel_Name"][:60], b["Kernel_Name"][:60]))
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"launches {len(rows)}  span {span/1e6:.1f} ms  kernel busy {busy/1e6:.1f} ms  idle between kernels {gap/1e6:.1f} ms ({100*gap/span:.1f} %)")
for k in ("<1us", "<2us", "<4us", "<10us", "<100us", ">=100us"):
    if k in hist:
        print(f"  gaps {k:7s}: {hist[k][0]:6d}  total {hist[k][1]/1e6:7.2f} ms")
bypair = defaultdict(lambda: [0, 0.0])
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g > 0:
        import re
        sh = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("ace355::(anonymous namespace)::", ""))[:34]
        k = sh(a["Kernel_Name"]) + " -> " + sh(b["Kernel_Name"])
        bypair[k][0] += 1
        bypair[k][1] += g
for k, (n, t) in sorted(bypair.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t/1e6:7.2f} ms  {n:5d} x {t/n/1e3:6.2f} us  {k}")
for g, x, y in sorted(big, reverse=True)[:4]:
    print(f"  {g/1e3:9.1f} us between {x} -> {y}")
