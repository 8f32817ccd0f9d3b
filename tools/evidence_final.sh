#!/bin/bash
# Round-end evidence on ONE box, for the commit given: the whole GPU suite + smoke, the default bench line, the
# rocprofv3 kernel-trace stats of the bench command, the FETCH_SIZE / WRITE_SIZE / MFMA-busy PMC passes (each its own run, no tracing
# domains beside the counters), the per-batch lines.  Every PMC summary is stamped with the library's source sha and the commit
# (tools/pmc_summary.py), which is what bench.py matches `roofline.traffic` against.
# Usage (from the build container): gpurun -- "bash tools/evidence_final.sh r04_final $(git rev-parse --short HEAD)"
TAG=${1:-r04_final}
export ACE355_COMMIT=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
echo "# commit $ACE355_COMMIT, lib_src_sha $(python -c 'import bench; print(bench.library_source_sha())')" > $OUT/${TAG}_gpu_pytest_measured.log
timeout 3000 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_gpu_pytest_measured.log
tail -3 $OUT/${TAG}_gpu_pytest_measured.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/${TAG}_gpu_pytest_measured.log
bash tools/profile_round.sh $TAG
bash tools/mfma_busy_pass.sh $TAG
# (after the PMC summaries exist in gpurun_out/: copy them where bench.py looks, so the line below carries its own traffic figure)
cp $OUT/${TAG}_pmc_FETCH_SIZE.json $OUT/${TAG}_pmc_WRITE_SIZE.json $ROOT/profiles/ 2>/dev/null
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line.json
python -c "
import json; d=json.load(open('$OUT/${TAG}_bench_line.json')); r=d['roofline']
print('bench', round(d['value'],3), 'songs/s', round(d['ms_per_step'],1), 'ms; gemm frac', round(r['frac'],4), 'traffic', r.get('traffic'), 'cpu', d.get('cpu_baseline',{}).get('value'))"
for B in 1 2 3 4; do
  python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --batch $B 2>/dev/null | tail -1 > $OUT/${TAG}_b${B}_vae_line.json
  python -c "import json; d=json.load(open('$OUT/${TAG}_b${B}_vae_line.json')); print('batch $B with decode', round(d['ms_per_step'],2), 'ms', round(d['value'],3), 'chains', d['config'].get('sampler_chains_per_gpu'))"
done
python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 --batch 1 --no-vae 2>/dev/null | tail -1 > $OUT/${TAG}_b1_line.json
python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 5 --batch 1 --no-vae --duration 10 --infer-steps 10 2>/dev/null | tail -1 > $OUT/${TAG}_cfg0_line.json
python -c "
import json
for n in ('b1', 'cfg0'):
    d=json.load(open('$OUT/${TAG}_%s_line.json' % n)); print(n, round(d['ms_per_step'],2), 'ms')"
