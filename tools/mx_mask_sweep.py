#!/usr/bin/env python
"""MXFP8 subsets: forward error vs the reference golden G11 and the forward time at the metric shape for ONE value of ACE355_MX_MASK
(1 qkv, 2 o_proj, 4 gate|up, 8 down; the library reads the variable once per process: run once per mask).  Usage: mx_mask_sweep.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native, weightgen
from ace355.dit import NativeDit, prepare_noise
dev = torch.device("cuda:0")
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
cfg = ace355.DitConfig()
dit = NativeDit(cfg, dev)
for name, shape in cfg.weight_shapes().items():
    wt = weightgen.make_dit_weights({name: shape}, cfg.hidden_size, seed=4, mode="test")[name]
    native.check(dit._lib.ace355_dit_load_tensor(dit._h, name.encode(), native.ptr(wt.contiguous()), 0, wt.numel(), 0), name)
native.check(dit._lib.ace355_dit_finalize(dit._h), "finalize")
null = weightgen.make_null_condition_emb(cfg.hidden_size, seed=4)
G = np.load(f"{gd}/g11_metric_forward.npz")
enc = torch.from_numpy(np.load(f"{gd}/g4_full_forward.npz")["enc"])
B, T = 8, 750
x8 = prepare_noise((B, T, 64), [1000 + i for i in range(B)])
g = torch.Generator().manual_seed(45)
ctx1 = torch.cat([0.5 * torch.randn(1, T, 64, generator=g), torch.ones(1, T, 64)], -1)
dit.set_condition(0, enc[0]); dit.set_condition(1, null.reshape(1, -1), L=enc.shape[1])
t = [float(G["t"])] * (2 * B)
args = (torch.cat([x8, x8]).to(dev), ctx1.expand(2 * B, -1, -1).contiguous().to(dev), t, t, [0] * B + [1] * B)
ref = torch.from_numpy(G["v"])
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
def timed():
    for _ in range(3): dit.forward(*args)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): v = dit.forward(*args)
    torch.cuda.synchronize()
    return v, (time.perf_counter() - t0) / 10 * 1e3
vb, tb = timed()
dit.set_precision("mxfp8")
vm, tm = timed()
print(f"mask {os.environ.get('ACE355_MX_MASK', '15(default: all four)')}: forward {tm:.2f} ms (bf16 {tb:.2f} ms), rel L2 vs reference fp32 {rel(vm.cpu(), ref):.3e} (bf16 {rel(vb.cpu(), ref):.3e})")
