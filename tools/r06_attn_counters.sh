#!/bin/bash
# Round 6, verdict item N3 (attention inner loop): hardware counters of the two attention kernels of the metric pass, each group in its
# own rocprofv3 pass (counters only, no tracing domains), restricted to the attention kernels.  What the passes answer:
#   A  how busy the matrix pipe and the VALU are while an attention kernel runs (SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU vs SQ_BUSY_CYCLES)
#   B  whether the fragment reads conflict in LDS (SQ_LDS_BANK_CONFLICT vs SQ_LDS_IDX_ACTIVE)
#   C  instruction mix per wave (SQ_INSTS_VALU, SQ_INSTS_MFMA / SQ_INSTS_VALU_MFMA_MOPS as available, SQ_INSTS_LDS, SQ_WAVES)
# Usage: gpurun -- "bash tools/r06_attn_counters.sh"  -> gpurun_out/r06_attn_counters.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
REP=$OUT/r06_attn_counters.txt
echo "# attention kernels of one metric pass (bench.py --steps 1 --warmup 0), lib_src_sha $(cd $ROOT && python -c 'import bench; print(bench.library_source_sha())')" > $REP
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > /tmp/avail.txt
echo "# counters named SQ_* on this box: $(wc -l < /tmp/avail.txt)" >> $REP
pass() {  # name, counters...
  local name=$1; shift
  local want=() c
  for c in "$@"; do if grep -qx "$c" /tmp/avail.txt || [[ $c != SQ_* ]]; then want+=($c); else echo "# pass $name: $c not offered here" >> $REP; fi; done
  rm -rf /tmp/prof_$name
  timeout 420 rocprofv3 --pmc "${want[@]}" --kernel-include-regex "attn_gqa_kernel|attn_rot_kernel" --output-format csv -d /tmp/prof_$name -- \
    python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /tmp/pass_$name.log 2>&1
  echo "# pass $name: ${want[*]} (rocprofv3 rc=$?)" >> $REP
  python - /tmp/prof_$name >> $REP <<'EOF'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set); dur = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "attn_gqa_kernel<6>" if "attn_gqa_kernel<6>" in k else "attn_rot_kernel<3>" if "attn_rot_kernel<3>" in k else k[:60]
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"]); n[fam].add(r["Dispatch_Id"])
        dur[fam][r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for fam in sorted(agg):
    L = len(n[fam])
    print(f"{fam}: {L} launches, {sum(dur[fam].values()) / L / 1e3:.1f} us per launch in this pass; per launch: " +
          ", ".join(f"{c} {v / L:.4g}" for c, v in sorted(agg[fam].items())))
EOF
}
pass A SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass B SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
pass C SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVES SQ_WAIT_INST_ANY
pass D SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
cat $REP
