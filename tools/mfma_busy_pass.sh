#!/bin/bash
# MFMA utilisation from hardware counters: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in their own rocprofv3
# pass (no tracing domains), restricted to the GEMM kernels (the unrestricted pass hung / crashed the tool in round 1).
# Usage: bash tools/mfma_busy_pass.sh TAG  -> gpurun_out/${TAG}_pmc_MFMA.json (+ .log)
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-include-regex "gemm_sp_kernel" --output-format csv -d /tmp/prof_mfma -- \
  python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $OUT/${TAG}_pmc_MFMA.log 2>&1
echo "rocprofv3 rc=$?" >> $OUT/${TAG}_pmc_MFMA.log
python $ROOT/tools/pmc_summary.py /tmp/prof_mfma $OUT/${TAG}_pmc_MFMA.json
