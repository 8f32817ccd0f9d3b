#!/bin/bash
# Round 5: ablations of attn_rot_kernel at the metric's cross-attention shape (ACE355_ATTN_CLK bits: 2 softmax off, 4 no DMA, 8 no barrier, 16 no MFMA)
cd $GRAFT_REPO_ROOT
# the ablation instantiations live in a second build of the library (attn.hip with -DACE355_ATTN_ABL; tools/_ab/libace355_abl.so)
LIB=ace-step-1.5-for-windows_amd/csrc/libace355.so
cp $LIB /tmp/_prod.so; cp tools/_ab/libace355_abl.so $LIB
trap "cp /tmp/_prod.so $LIB" EXIT
cat > /tmp/abl.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
N, Sq, Skv, win = [int(x) for x in sys.argv[1:5]]
g = torch.Generator(device=dev).manual_seed(1)
q = torch.randn(N, Sq, 2048, device=dev, generator=g).to(torch.bfloat16); k = torch.randn(N, Skv, 1024, device=dev, generator=g).to(torch.bfloat16)
v = torch.randn(N, Skv, 1024, device=dev, generator=g).to(torch.bfloat16); o = torch.empty_like(q)
for _ in range(6):
    native.check(lib.ace355_attention(P(q), P(k), P(v), P(o), N, Sq, Skv, 16, 8, win, 128 ** -0.5, None))
torch.cuda.synchronize()
PY
for shape in "8 375 769 -1"; do
for rot in 1; do
for f in 1 3 5 9 17 13 15 29 31; do
  echo "shape $shape ROT=$rot flags=$f: $(ACE355_ATTN_ROT=$rot ACE355_ATTN_CLK=$f python /tmp/abl.py $shape 2>&1 | grep 'attn-rot clk' | tail -3 | sed -e 's/.*GHz, //' | tr '\n' '|')"
done; done; done
