#!/usr/bin/env python
"""Turbo-variant timing (8-step tables, no CFG) at the headline shape: B x 30 s requests, DiT sampler + VAE decode + peak
normalise.  Not the BASELINE metric (that is the base/sft 27-step CFG path, bench.py); this is the DESIGN.md row for the
turbo model family, which shares every kernel with it.
Usage: python tools/bench_turbo.py [--batch 8] [--seconds 30] [--shift 3.0] [--iters 5]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (GPU-side synthetic weights of the headline bench)
from ace355.dit import generate_latents_turbo  # noqa: E402
from ace355.vae import peak_normalize  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--shift", type=float, default=3.0)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--enc-len", type=int, default=512)
    args = ap.parse_args()
    dev = "cuda:0"
    B, T, L = args.batch, int(round(args.seconds * 25)), args.enc_len
    args.tiny, args.no_vae = False, False
    cfg, vcfg, dit, vae, _, _ = bench.build_models(args, torch.device(dev))
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(1, L, cfg.hidden_size, generator=g).expand(B, -1, -1).contiguous()
    ctx = torch.randn(1, T, cfg.in_channels - cfg.audio_acoustic_hidden_dim, generator=g).expand(B, -1, -1).contiguous()
    noise = torch.randn(B, T, cfg.audio_acoustic_hidden_dim, generator=g)

    def one():
        lat = generate_latents_turbo(dit, enc, ctx, shift=args.shift, noise=noise)["target_latents"]
        return peak_normalize(vae.decode(lat.transpose(1, 2).contiguous()))

    for _ in range(2):
        out = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        out = one()
    torch.cuda.synchronize()
    s = (time.perf_counter() - t0) / args.iters
    assert torch.isfinite(out).all()
    print(json.dumps({"workload": f"turbo: {B} x {args.seconds:g} s, 8 steps (shift {args.shift:g}), no CFG, DiT + VAE decode",
                      "ms_per_pass": round(s * 1e3, 2), "songs_per_s": round(B / s, 2)}))


if __name__ == "__main__":
    main()
