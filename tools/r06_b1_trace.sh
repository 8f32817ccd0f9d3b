#!/bin/bash
# One song per request (configs[1] / the 8-GPU share of the metric batch): kernel trace of the DiT-only bench, reduced to (a) the per-kernel
# totals of one pass and (b) the launch sequence of ONE decoder layer in the middle of a forward with start-to-start gaps
# -> gpurun_out/r06_b1_trace.txt.   Usage: gpurun -- "bash tools/r06_b1_trace.sh [BATCH]"
B=${1:-1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b1
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b1 -- python $ROOT/bench.py --batch $B --no-vae --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/b1.log 2>&1
tail -1 /tmp/b1.log | cut -c1-300
python - /tmp/prof_b1 $B > $OUT/r06_b${B}_trace.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("ace355::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:70]
# the last pass = after the last apg/euler... simply: take the last 27 * 24 layers worth; find layer boundaries by the QKV head-norm GEMM (mode 4 with N=4096: first mode-4 gemm of a layer)
names = [short(r["Kernel_Name"]) for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
# totals over the LAST timed pass: find the step kernel (apg_euler) occurrences; a pass has 27 of them
idx = [i for i, n in enumerate(names) if "apg_euler" in n or "euler" in n]
first = idx[-27] if len(idx) >= 27 else 0
prev = idx[-28] if len(idx) >= 28 else 0
lo, hi = prev + 1, idx[-1] + 1
tot = collections.defaultdict(lambda: [0, 0.0])
for i in range(lo, hi):
    tot[names[i]][0] += 1; tot[names[i]][1] += (en[i] - st[i]) / 1e3
wall = (en[hi - 1] - st[lo]) / 1e3
busy = sum(v[1] for v in tot.values())
print(f"# batch {sys.argv[2]}, DiT only: last pass = {hi - lo} launches, wall {wall / 1e3:.2f} ms, kernel time {busy / 1e3:.2f} ms, idle between kernels {100 * (1 - busy / wall):.1f} %")
for n, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{t / 1e3:8.2f} ms {c:6d} x {t / c:7.1f} us  {n}")
# one forward in the middle: between two consecutive step kernels; one layer in its middle
a, b = idx[-14] + 1, idx[-13]
seq = list(range(a, b))
print(f"# one forward: {len(seq)} launches, {(en[b - 1] - st[a]) / 1e3:.1f} us")
n_layer = len(seq) // 24
mid = a + (len(seq) // 2 // max(1, n_layer)) * n_layer
# print ~1.5 layers of launches with gaps
for i in range(mid, min(b, mid + int(1.6 * n_layer) + 2)):
    g = r = rows[i]
    print(f"{(st[i] - st[a]) / 1e3:9.1f} us  dur {(en[i] - st[i]) / 1e3:6.1f}  gap_before {(st[i] - en[i - 1]) / 1e3:5.1f}  grid {int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])):5d}x{int(r['Grid_Size_Y']):3d} wg {r['Workgroup_Size_X']:>4s}  {names[i]}")
PY
cat $OUT/r06_b${B}_trace.txt | head -80
