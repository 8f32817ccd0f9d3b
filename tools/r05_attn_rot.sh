#!/bin/bash
# Round 5: the rotated key-split attention kernel (ACE355_ATTN_ROT = 0 off | 1 where the 6-wave GQA kernel ran | 2 wherever a GQA kernel ran):
# parity (tests/test_kernels_gpu.py::test_attention + the masked variant) per setting, then the probe's four shapes under the kernel tracer.
cd $GRAFT_REPO_ROOT
for st in 1 2; do
  echo "== parity ACE355_ATTN_ROT=$st"
  ACE355_ATTN_ROT=$st timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_cond_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -4
done
cd /tmp && export TMPDIR=/tmp
for st in 0 1 2; do echo "== timing ACE355_ATTN_ROT=$st"; rm -rf /tmp/at$st; ACE355_ATTN_ROT=$st rocprofv3 --kernel-trace --output-format csv -d /tmp/at$st -- python $GRAFT_REPO_ROOT/tools/attn_probe.py 2>&1 | grep "rel L2"; python - /tmp/at$st <<PY
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"] and "merge" not in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
for g in range(0, len(rows), 8):
    grp = rows[g:g + 8]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp[2:])
    name = grp[0]["Kernel_Name"].split("(")[0].split("::")[-1].replace("void ", "")
    print(f"  shape {g // 8}: {name} grid {int(grp[0]['Grid_Size_X']) // int(grp[0]['Workgroup_Size_X'])} x {grp[0]['Workgroup_Size_X']}: median {d[len(d) // 2]:.1f} us")
PY
done
for st in 0 1 2; do ACE355_ATTN_ROT=$st ACE355_ATTN_CLK=1 python $GRAFT_REPO_ROOT/tools/attn_probe.py 2>&1 | grep "clk" | sort | uniq -c | sort -rn | head -8; done
