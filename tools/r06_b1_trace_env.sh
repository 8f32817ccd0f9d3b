#!/bin/bash
# tools/r06_b1_trace.sh with the environment passed through (A/B of launch-level switches at one song): per-kernel averages of the last pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b1e
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b1e -- python $ROOT/bench.py --batch 1 --no-vae --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/b1e.log 2>&1
python - /tmp/prof_b1e <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("void ", "").replace("ace355::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
idx = [i for i, r in enumerate(rows) if "apg_euler" in r["Kernel_Name"]]
a, b = idx[-14] + 1, idx[-13]
n_layer = (b - a) // 24
mid = a + ((b - a) // 2 // n_layer) * n_layer
for r in rows[mid:mid + n_layer + 1]:
    print(f"dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f}  grid {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):4d}x{r['Grid_Size_Y']:>2s}  {short(r['Kernel_Name'])}")
print(f"one forward {(int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us")
PY
