#!/usr/bin/env python
"""Decode (and encode) one fixed batch through the library in csrc/ and print checksums + timings: run once per build (tools/ab_lib.sh
pattern) to show two builds produce bit-identical waveforms."""
import os, sys, time, hashlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355.vae import NativeVae
from ace355.config import VaeConfig
from ace355 import weightgen
dev = torch.device("cuda:0")
cfg = VaeConfig()
vae = NativeVae(cfg, dev)
vae.load_state_dict(weightgen.make_vae_weights({**cfg.weight_shapes(), **cfg.encoder_weight_shapes()}, seed=4, mode="init"))
B, T = int(os.environ.get("VB", "8")), int(os.environ.get("VT", "750"))
g = torch.Generator().manual_seed(3)
z = torch.randn(B, 64, T, generator=g).to(dev)
for _ in range(2):
    w = vae.decode(z)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    w = vae.decode(z)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
h = hashlib.sha256(w.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"decode B={B} T={T}: {dt*1e3:.2f} ms  sha {h}  absmax {float(w.abs().max()):.4f}", flush=True)
a = torch.randn(2, 2, 48000 * 10, generator=g).to(dev) * 0.1
lat = vae.encode(a, sample=False)
torch.cuda.synchronize()
print(f"encode 2 x 10 s: sha {hashlib.sha256(lat.float().cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
