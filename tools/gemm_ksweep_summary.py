#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace of tools/gemm_ksweep_trace.py: median of the last 4 of each 6 consecutive gemm launches."""
import csv, glob, sys
root = sys.argv[1]; shapes = sys.argv[2:]
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i, sh in enumerate(shapes):
    grp = rows[6 * i + 2: 6 * i + 6]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp)
    M, N, K = (int(v) for v in sh.split(","))
    med = d[len(d) // 2]
    print(f"{sh:18s} {med:8.1f} us  {2.0*M*N*K/med/1e6:7.1f} TF/s  grid={grp[0]['Grid_Size_X']}")
