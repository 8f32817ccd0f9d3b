#!/bin/bash
# per-launch durations of one decode with every unfused launch forced onto the 4-wave (128) / 8-wave (256) tile, at 1 / 2 / 4 songs
# (after conv.hip version 2: the tile-height rule of round 4 was measured on a kernel whose 8-wave form spilled)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for nb in 1 2 4; do
for tm in 128 256; do
  echo "== B=$nb ACE355_CONV_TM=$tm"
  rm -rf /tmp/ct_$tm
  B=$nb ACE355_CONV_TM=$tm rocprofv3 --kernel-trace --output-format csv -d /tmp/ct_$tm -- python tools/vae_trace.py > /dev/null 2>&1
  python tools/vae_trace_list.py /tmp/ct_$tm | tail -34
done
done
