#!/usr/bin/env python
"""Decode one batch with the native VAE (run under rocprofv3 --kernel-trace to get per-conv durations)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import weightgen
from ace355.vae import NativeVae
B, T = int(os.environ.get("B", 8)), int(os.environ.get("T", 750))
cfg = ace355.VaeConfig()
vae = NativeVae(cfg, "cuda:0")
vae.load_state_dict(weightgen.make_vae_weights(cfg.weight_shapes(), seed=1, mode="init"))
z = torch.randn(B, 64, T, device="cuda:0")
for _ in range(2):
    w = vae.decode(z)
torch.cuda.synchronize()
print("ok", w.shape)
