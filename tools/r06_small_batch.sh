#!/bin/bash
# Round 6: the small requests (1 / 2 songs of 30 s: the 8- and 4-GPU shares of the metric batch) under the switches that exist:
# hipGraph replay of the sampler, the per-layer CFG fork, one vs two chains.  One box, interleaved (ABAB), DiT + decode, ms per request.
set -u
cd "$(dirname "$0")/.."
line() { python bench.py --no-cpu-baseline --no-roofline --batch $1 --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"; }
for rep in 1 2; do
  for b in 1 2; do
    echo "b=$b default            $(line $b)"
    echo "b=$b graph              $(ACE355_SAMPLE_GRAPH=1 line $b)"
    echo "b=$b fork2 (one chain)  $(ACE355_CFG_FORK=2 ACE355_DUAL=0 line $b)"
    echo "b=$b one chain          $(ACE355_DUAL=0 line $b)"
    echo "b=$b graph one chain    $(ACE355_SAMPLE_GRAPH=1 ACE355_DUAL=0 line $b)"
  done
done
