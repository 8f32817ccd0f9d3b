#!/bin/bash
# Round 6 ("GEMM traffic 2.3-2.5 x algorithmic ... answered with a model rather than an experiment"): the fabric-side FETCH bytes per
# GEMM launch (rocprofv3 --pmc FETCH_SIZE, its own pass, GEMM kernels only) and the time per 8-song pass under the rasterisation knobs that exist:
# group height inside an XCD region (ACE355_GEMM_GROUPM, default 4) and the XCD grid shape (ACE355_GEMM_XCDM, default: min xn |A| + xm |W|).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "A=0" "ACE355_GEMM_GROUPM=2" "ACE355_GEMM_GROUPM=8" "ACE355_GEMM_GROUPM=16" "ACE355_GEMM_XCDM=8" "ACE355_GEMM_XCDM=2"; do
  rm -rf /tmp/prof_r
  env $v timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gemm_sp_kernel" --output-format csv -d /tmp/prof_r -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/prof_r.log 2>&1
  python $ROOT/tools/pmc_summary.py /tmp/prof_r /tmp/prof_r.json > /dev/null 2>&1
  F=$(python -c "import json; d=json.load(open('/tmp/prof_r.json')); print('%.1f MB x 2 (gfx950 correction) per launch over %d launches' % (d['gemm']['FETCH_SIZE']['per_launch']*1024/1e6, d['gemm']['FETCH_SIZE']['launches']))")
  T=$(env $v python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f ms per pass, GEMM %.4f' % (d['ms_per_step'], d['roofline']['frac']))")
  echo "$v: FETCH $F; $T"
done
