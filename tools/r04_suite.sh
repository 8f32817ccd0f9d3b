#!/bin/bash
# Round 4: the whole GPU suite + smoke + the per-batch bench lines on one box.  bash tools/r04_suite.sh TAG
TAG=${1:-r04_v8}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 3000 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_gpu_pytest.log
tail -5 $OUT/${TAG}_gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT/${TAG}_gpu_pytest.log
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line.json
python -c "import json; d=json.load(open('$OUT/${TAG}_bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('sampler_chains_per_gpu'))"
for B in 1 2 4; do
  python bench.py --no-cpu-baseline --steps 10 --warmup 3 --batch $B 2>/dev/null | tail -1 > $OUT/${TAG}_b${B}_vae_line.json
  python -c "import json; d=json.load(open('$OUT/${TAG}_b${B}_vae_line.json')); print('batch $B with decode', round(d['ms_per_step'],2), 'ms', round(d['value'],3), d['config'].get('sampler_chains_per_gpu'))"
done
python bench.py --no-cpu-baseline --steps 10 --warmup 3 --batch 1 --no-vae 2>/dev/null | tail -1 > $OUT/${TAG}_b1_line.json
python -c "import json; d=json.load(open('$OUT/${TAG}_b1_line.json')); print('batch 1 DiT only', round(d['ms_per_step'],2))"
python bench.py --no-cpu-baseline --steps 20 --warmup 5 --batch 1 --no-vae --duration 10 --infer-steps 10 2>/dev/null | tail -1 > $OUT/${TAG}_cfg0_line.json
python -c "import json; d=json.load(open('$OUT/${TAG}_cfg0_line.json')); print('cfg0', round(d['ms_per_step'],2))"
