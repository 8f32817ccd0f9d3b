#!/usr/bin/env python
"""What would a bf16 residual stream cost in accuracy?  ("measure - do not assume".)  CPU only: the fp32 oracle
forward at the G4 shape (full architecture, N = 2, T = 250, L = 769, the fixture's inputs and weights; checked against the reference's
vector first) is re-run with the residual stream h rounded to bf16 after the patchify projection and after each of the 72 residual
adds - everything else stays fp32, so the number is the error the storage format ALONE adds on top of whatever the kernels do.
Usage: python tools/bf16_residual_estimate.py"""
import inspect, os, re, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355  # noqa: E402
from ace355 import weightgen  # noqa: E402
from oracle import dit as o_dit  # noqa: E402

torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g4_full_forward.npz"))
cfg = o_dit.DitConfig()
w = weightgen.make_dit_weights(ace355.DitConfig().weight_shapes(), cfg.hidden_size, seed=int(G["seed"]), mode="test")
null = weightgen.make_null_condition_emb(cfg.hidden_size, seed=int(G["seed"]))
x, ctx, enc, t = (torch.from_numpy(G[k]) for k in ("x", "ctx", "enc", "t"))
enc2 = torch.cat([enc, null.expand_as(enc)], 0)
ref = torch.from_numpy(G["v"])
rel = lambda a, b: float((a - b).norm() / b.norm())
taps0 = {}
with torch.no_grad():
    v0 = o_dit.dit_forward(cfg, w, x, t, t, enc2, ctx, o_dit.CrossCache(), taps0)
print(f"fp32 oracle vs the reference's vector (G4): {rel(v0, ref):.2e}")
src = inspect.getsource(o_dit.dit_layer)
src, n = re.subn(r"(\n    h = h \+ [^\n]+)", r"\1\n    h = h.to(torch.bfloat16).float()", src)
assert n == 3, n
ns = dict(o_dit.__dict__)
exec(src, ns)
fsrc = inspect.getsource(o_dit.dit_forward)
fsrc, n = re.subn(r"(\n    h = F\.conv1d\([^\n]+)", r"\1\n    h = h.to(torch.bfloat16).float()", fsrc)
assert n == 1, n
ns["dit_layer"] = ns["dit_layer"]
exec(fsrc, ns)
taps1 = {}
with torch.no_grad():
    v1 = ns["dit_forward"](cfg, w, x, t, t, enc2, ctx, o_dit.CrossCache(), taps1)
print(f"bf16-rounded residual stream vs fp32: velocity rel L2 {rel(v1, v0):.3e}; residual stream after layer 0 {rel(taps1['l0.out'], taps0['l0.out']):.3e}, "
      f"after layer 23 {rel(taps1['l23.out'], taps0['l23.out']):.3e}")
print("native path today (fp32 residual stream, bf16 GEMM operands) vs the reference at this shape: 5.8e-3 (gate 1.5e-2); taps 3.8e-3 / 4.1e-3 (gates 1e-2 / 1.2e-2)")
