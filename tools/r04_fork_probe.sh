#!/bin/bash
# Round 4, CFG fork: parity of the new paths, same-box ABAB of the fork and its knobs, and a kernel trace that shows whether the two
# streams overlapped.  One gpurun call: bash tools/r04_fork_probe.sh TAG
TAG=${1:-r04_v1}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "cfg_fork or unordered_split or metric_length or deep_window" 2>&1 | grep -v amdgpu.ids | tail -25 > $OUT/${TAG}_new_tests.log
tail -8 $OUT/${TAG}_new_tests.log
echo "== ABAB fork off / on" | tee $OUT/${TAG}_fork_ab.txt
bash tools/ab_env.sh ACE355_CFG_FORK=0 ACE355_CFG_FORK=1 2 -- --steps 6 --warmup 2 2>&1 | tee -a $OUT/${TAG}_fork_ab.txt
for V in "ACE355_FORK_DOWN_BIG=0" "ACE355_FORK_NOPERS=1" "ACE355_FORK_PRIO=1" "ACE355_FORK_PRIO=-1" "ACE355_CFG_FORK=0"; do
  echo "== $V" | tee -a $OUT/${TAG}_fork_ab.txt
  env $V python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', round(d['value'],3))" | tee -a $OUT/${TAG}_fork_ab.txt
done
cd /tmp && export TMPDIR=/tmp
for F in 1 0; do
  rm -rf /tmp/tr_$F
  ACE355_CFG_FORK=$F timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$F -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace_fork$F.log 2>&1
  python $ROOT/tools/overlap_summary.py /tmp/tr_$F > $OUT/${TAG}_overlap_fork$F.txt 2>&1
  head -3 $OUT/${TAG}_overlap_fork$F.txt
done
