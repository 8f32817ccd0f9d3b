#!/bin/bash
# Round 4: two half-batch chains side by side, launches shaped for the whole chip vs for half of it.  bash tools/r04_dual_probe.sh TAG
TAG=${1:-r04_v2}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python tools/dual_chain_probe.py --offset-ms 0 0.35 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_dual_256.txt
ACE355_MAX_WGS=128 python tools/dual_chain_probe.py --skip-single --offset-ms 0 0.35 2>&1 | grep -v amdgpu.ids | tee $OUT/${TAG}_dual_128.txt
cd /tmp && export TMPDIR=/tmp
for C in 128 256; do
rm -rf /tmp/tr_d
ACE355_MAX_WGS=$C timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_d -- python $ROOT/tools/dual_chain_probe.py --skip-single --offset-ms 0.35 --iters 1 > $OUT/${TAG}_trace_dual_$C.log 2>&1
python - <<'PY' > $OUT/${TAG}_dual_timeline_$C.txt 2>&1
import csv, glob, re, collections
f = glob.glob("/tmp/tr_d/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sh = lambda n: re.sub(r"\(.*", "", n.replace("void ", "").replace("ace355::(anonymous namespace)::", ""))[:44]
print("launches per queue:", dict(collections.Counter(r.get("Queue_Id", "?") for r in rows)))
n = len(rows)
win = rows[int(n * 0.5):]
ev = sorted([(int(r["Start_Timestamp"]), 1) for r in win] + [(int(r["End_Timestamp"]), -1) for r in win])
d = 0; last = ev[0][0]; ge1 = ge2 = 0
for t, x in ev:
    if d >= 1: ge1 += t - last
    if d >= 2: ge2 += t - last
    d += x; last = t
print(f"second half of the trace: >=1 kernel running {ge1/1e6:.1f} ms, >=2 running {ge2/1e6:.1f} ms")
seg = rows[int(n * 0.8): int(n * 0.8) + 50]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    g, w = r.get("Grid_Size") or r.get("Grid_Size_X") or "0", r.get("Workgroup_Size") or r.get("Workgroup_Size_X") or "1"
    print(f"{(int(r['Start_Timestamp']) - t0)/1e3:8.1f} -> {(int(r['End_Timestamp']) - t0)/1e3:8.1f} ({(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:7.1f} us) queue {r.get('Queue_Id','?'):>3} wgs {int(g)//max(1,int(w)):5d}  {sh(r['Kernel_Name'])}")
PY
head -34 $OUT/${TAG}_dual_timeline_$C.txt
done
