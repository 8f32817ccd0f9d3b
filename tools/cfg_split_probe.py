#!/usr/bin/env python
"""Feasibility of running the conditional and the null sequences of a small-batch CFG forward as two concurrent chains: one batch-B
CFG sampler (2B sequences per forward, one stream) against two guidance-1 samplers of B songs each on two streams (two handles; both
chains carry cross-attention here, so this bounds the split forward from above).  Usage: cfg_split_probe.py [--batch 1]"""
import argparse, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ace355.dit import SLOT_COND, SLOT_NULL, NativeDit, schedule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--steps", type=int, default=27)
ap.add_argument("--iters", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
args.tiny, args.no_vae, args.fp8 = False, True, False
dcfg, _, dit_a, _, sd, _ = bench.build_models(args, dev)
dit_b = NativeDit(dcfg, dev)
dit_b.load_state_dict(sd)
B, T, L = args.batch, 750, 769
g = torch.Generator().manual_seed(0)
enc = torch.randn(L, dcfg.hidden_size, generator=g).to(dev)
null = torch.randn(1, dcfg.hidden_size, generator=g).to(dev)
ctx = torch.randn(B, T, 128, generator=g).to(dev)
noise = torch.randn(B, T, 64, generator=g).to(dev)
ts = schedule(args.steps, 1.0, None)
for d in (dit_a, dit_b):
    d.set_condition(SLOT_COND, enc)
    d.set_condition(SLOT_NULL, null.reshape(1, -1), L=L)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

def single_cfg():
    return dit_a.sample(noise, ctx, ts, guidance_scale=7.0)

def single_nocfg():
    return dit_a.sample(noise, ctx, ts, guidance_scale=1.0)

def dual_nocfg():
    def work(i, d):
        with torch.cuda.stream(streams[i]):
            d.sample(noise, ctx, ts, guidance_scale=1.0)
            streams[i].synchronize()
    th = [threading.Thread(target=work, args=(i, d)) for i, d in enumerate((dit_a, dit_b))]
    for t in th: t.start()
    for t in th: t.join()

for name, fn in (("one CFG sampler (2B sequences)", single_cfg), ("one guidance-1 sampler (B sequences)", single_nocfg),
                 ("two guidance-1 samplers on two streams", dual_nocfg)) * 2:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / args.iters * 1e3:.1f} ms per pass (batch {B})", flush=True)
