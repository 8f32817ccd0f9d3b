#!/bin/bash
# timeline of a few layers with the prefetch on: does prefetch_kernel overlap the GEMMs, and what does it do to their durations?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pf
ACE355_PREFETCH=1 ACE355_PREFETCH_SYNC=${SY:-0} ACE355_PREFETCH_SPLIT=${SP:-1} ACE355_PREFETCH_BS=64 ACE355_PREFETCH_UNROLL=2 ACE355_PREFETCH_WGS=${WG:-256} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_pf -- python $ROOT/bench.py --batch 1 --no-vae --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pf.log 2>&1
python - /tmp/prof_pf <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
def short(n):
    return n.replace("void ", "").replace("ace355::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
pf = [i for i, r in enumerate(rows) if "prefetch_kernel" in r["Kernel_Name"]]
print("prefetch launches", len(pf))
i0 = pf[len(pf) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 3:i0 + 30]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  q {r.get('Queue_Id', '?'):>3s}  {short(r['Kernel_Name'])}")
tot = collections.defaultdict(lambda: [0, 0.0])
for r in rows[len(rows) // 2:]:
    tot[short(r["Kernel_Name"])][0] += 1; tot[short(r["Kernel_Name"])][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:9]:
    print(f"{t / c:8.1f} us avg x {c:5d}  {n}")
PY
