#!/usr/bin/env python
"""Did two streams really overlap?  From a rocprofv3 kernel trace of bench.py: over one steady-state pass, the sum of kernel durations,
the union of their intervals (time at least one kernel is running) and the time at least TWO kernels are running; then one decoder
layer's launches as a timeline (start / end relative to the layer's QKV GEMM, queue id, workgroups) so the fork can be read off.
Usage: rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py ...; python tools/overlap_summary.py DIR [layer_index]"""
import csv, glob, re, sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "peak_scale_kernel" in r["Kernel_Name"]]
if len(marks) >= 2:
    rows = rows[marks[-2] + 1:marks[-1] + 1]     # the last whole pass
sh = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("ace355::(anonymous namespace)::", ""))[:44]
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth = 0
last = ev[0][0]
ge1 = ge2 = 0
for t, d in ev:
    if depth >= 1: ge1 += t - last
    if depth >= 2: ge2 += t - last
    depth += d
    last = t
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"pass: {len(rows)} launches, span {span/1e6:.2f} ms, sum of kernel durations {tot/1e6:.2f} ms, >=1 kernel running {ge1/1e6:.2f} ms, "
      f">=2 kernels running {ge2/1e6:.2f} ms ({100.0*ge2/span:.1f} % of the span)")
# per kernel name: calls, mean duration
agg = {}
for r in rows:
    k = sh(r["Kernel_Name"])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"  {t/1e6:8.2f} ms {n:6d} x {t/n/1e3:8.2f} us  {k}")
# one layer's timeline: the launches between two consecutive head-epilogue QKV GEMMs (gemm_sp_kernel<4, 3, 4, 2, 1 = the persistent mode-4 tile)
qkv = [i for i, r in enumerate(rows) if "gemm_sp_kernel<4, 3, 4, 2, 1" in r["Kernel_Name"]]
li = int(sys.argv[2]) if len(sys.argv) > 2 else len(qkv) // 2
if len(qkv) > li + 1:
    seg = rows[qkv[li]:qkv[li + 1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    print(f"layer timeline (launches {qkv[li]}..{qkv[li + 1]} of the pass; us from the QKV GEMM's start):")
    for r in seg:
        g = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        w = r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "?"
        try:
            wg = int(g) // max(1, int(w))
        except Exception:
            wg = "?"
        print(f"  {(int(r['Start_Timestamp']) - t0)/1e3:8.1f} -> {(int(r['End_Timestamp']) - t0)/1e3:8.1f}  ({(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:7.1f} us)  "
              f"queue {r.get('Queue_Id', '?'):>3}  wgs {wg!s:>5}  {sh(r['Kernel_Name'])}")
