#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: per-kernel-family totals and per-launch averages."""
import collections, csv, glob, json, sys
root, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
seen_d = set()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "gemm" if "gemm_" in k else "attn" if "attn" in k else "conv" if "conv_kernel" in k else "rmsnorm" if "rmsnorm" in k else "other"
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[(fam, r["Counter_Name"])].add(r["Dispatch_Id"])
        if "Start_Timestamp" in r and "End_Timestamp" in r and (fam, r["Dispatch_Id"]) not in seen_d:  # kernel time IN the profiled pass
            seen_d.add((fam, r["Dispatch_Id"]))
            agg[fam]["_duration_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            launches[(fam, "_duration_ns")].add(r["Dispatch_Id"])
res = {}
for fam, cs in agg.items():
    res[fam] = {c: {"total": v, "launches": len(launches[(fam, c)]), "per_launch": v / max(1, len(launches[(fam, c)]))} for c, v in cs.items()}
# provenance: which library the pass ran on (sha of its sources: there is no .git on the GPU box) and, when the caller knows it
# (ACE355_COMMIT, set from `git rev-parse --short HEAD` in the gpurun command line), the commit
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from bench import library_source_sha
    res["_meta"] = {"lib_src_sha": library_source_sha(), "commit": os.environ.get("ACE355_COMMIT") or None}
except Exception as e:  # (never lose a profile over its stamp)
    res["_meta"] = {"error": str(e)[:200]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({f: {c: round(d["per_launch"], 1) for c, d in cs.items()} for f, cs in res.items() if f != "_meta"}))
