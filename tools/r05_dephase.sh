#!/bin/bash
# ACE355_CONV_DEPHASE sweep: 8 x 30 s decode (and one song), sha + time per setting, ABAB-style (0 between the others)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp
for r in 1 2; do for d in 0 1 2 0 3 4; do
  echo "DEPHASE=$d: $(ACE355_CONV_DEPHASE=$d python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | head -1)"
done; done
for d in 0 2 0 2; do echo "one song DEPHASE=$d: $(ACE355_CONV_DEPHASE=$d VB=1 python tools/vae_ab_check.py 2>&1 | grep -v amdgpu.ids | head -1)"; done
