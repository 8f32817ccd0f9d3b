#!/usr/bin/env python
"""VAE encoder timing: B x 30 s of 48 kHz stereo -> [B, 64, 750] latents, with the achieved conv
TFLOP/s from the library's HIP-event profile and the fp32 oracle on a bounded sample (CPU, same box).
Usage: python tests/perf/bench_vae_encode.py [--batch 8] [--seconds 30] [--no-cpu]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ace355  # noqa: E402
from ace355 import weightgen  # noqa: E402
from ace355.vae import NativeVae  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    cfg = ace355.VaeConfig()
    w = weightgen.make_vae_weights({**cfg.weight_shapes(), **cfg.encoder_weight_shapes()}, seed=1, mode="init")
    vae = NativeVae(cfg, "cuda:0")
    vae.load_state_dict(w)
    frames = int(round(args.seconds * 25))
    L = cfg.hop * frames
    audio = (0.3 * torch.randn(args.batch, 2, L, generator=torch.Generator().manual_seed(0))).cuda()
    for _ in range(2):
        z = vae.encode(audio)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        z = vae.encode(audio)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    vae.set_profile(True)
    vae.encode(audio)
    torch.cuda.synchronize()
    p = vae.get_profile()
    vae.set_profile(False)
    out = {"workload": f"VAE encode, B={args.batch} x {args.seconds:g} s stereo 48 kHz -> {tuple(z.shape)}", "gpu_ms": round(ms, 2),
           "conv_tflop": round(p["conv_flops"] / 1e12, 2), "conv_tflops": round(p["conv_flops"] / (p["conv_ms"] * 1e-3) / 1e12, 1),
           "audio_s_per_s": round(args.batch * args.seconds / (ms * 1e-3), 1)}
    if not args.no_cpu:
        from oracle import oobleck as o_vae
        n = max(1, min(16, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n)
        fr = 12
        a = audio[:1, :, : cfg.hop * fr].cpu()
        t0 = time.perf_counter()
        ref, _ = o_vae.encode_moments(o_vae.VaeConfig(), w, a)
        cpu_s = time.perf_counter() - t0
        got = vae.encode(audio[:1, :, : cfg.hop * fr].contiguous(), sample=False).cpu()
        snr = float(10 * torch.log10(ref.pow(2).sum() / (got - ref).pow(2).sum()))
        out.update({"cpu_oracle_s_per_song": round(cpu_s * frames / fr, 2), "cpu_threads": n, "snr_db_vs_oracle": round(snr, 1),
                    "speedup_vs_cpu_oracle": round(cpu_s * frames / fr * args.batch / (ms * 1e-3), 1)})
    print(out)


if __name__ == "__main__":
    main()
