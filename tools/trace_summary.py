#!/usr/bin/env python
"""Print per-launch durations of kernels matching a pattern from a rocprofv3 kernel-trace CSV (second half = warm run)."""
import csv, glob, sys
pat, root = sys.argv[1], sys.argv[2]
names = sys.argv[3].split(",") if len(sys.argv) > 3 else None
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows = rows[len(rows) // 2:]
tot = 0.0
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    nm = names[i] if names and i < len(names) else str(i)
    print(f"{nm:10s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) {d:9.1f} us")
print("total", tot / 1e3, "ms")
