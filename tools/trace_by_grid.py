#!/usr/bin/env python
"""Group a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): launches, average / total time.  Usage: trace_by_grid.py DIR [pattern]"""
import collections, csv, glob, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if pat not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("void ace355::(anonymous namespace)::", "").replace("ace355::(anonymous namespace)::", "").replace("void ace355::", "").split("(")[0]
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += d
for (name, gx, gy, gz, wx), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{tot/1e3:9.2f} ms {n:6d} x {tot/n:8.1f} us  grid=({gx},{gy},{gz}) wg={wx}  {name[:90]}")
