#!/usr/bin/env python
"""Condition-encoder timing at the product shape: B items x (256 text + 512 lyric tokens) + one 30 s
reference clip (750 x 64 latent frames) per item.  Prints GPU ms per encode, achieved TFLOP/s, and the fp32 CPU oracle on
a bounded sample of the same workload (one item) for the same box.
Usage: python tests/perf/bench_cond.py [--batch 8] [--no-cpu]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ace355  # noqa: E402
from ace355 import weightgen  # noqa: E402
from ace355.cond import NativeCondEncoder  # noqa: E402


def flops(cfg, B, Lt, Ll, Nref, Tref):
    D, Fh = cfg.hidden_size, cfg.intermediate_size
    q, kv = cfg.num_attention_heads * cfg.head_dim, cfg.num_key_value_heads * cfg.head_dim
    per_tok_layer = 2 * (D * (q + 2 * kv) + q * D + 3 * D * Fh)
    def stack(n_layers, N, S, din):
        attn = 0
        for li in range(n_layers):
            keys = S if cfg.layer_types[li] == "full_attention" else min(S, 2 * cfg.sliding_window + 1)
            attn += 4 * N * cfg.num_attention_heads * S * keys * cfg.head_dim
        return N * S * (n_layers * per_tok_layer + 2 * din * D) + attn
    return (2 * B * Lt * cfg.text_hidden_dim * D + stack(cfg.num_lyric_encoder_hidden_layers, B, Ll, cfg.text_hidden_dim)
            + stack(cfg.num_timbre_encoder_hidden_layers, Nref, Tref, cfg.timbre_hidden_dim))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    cfg = ace355.CondConfig()
    B, Lt, Ll, Tref = args.batch, 256, 512, 750
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=3, mode="init")
    enc = NativeCondEncoder(cfg, "cuda:0")
    enc.load_state_dict(w)
    g = torch.Generator().manual_seed(0)
    text = torch.randn(B, Lt, cfg.text_hidden_dim, generator=g).cuda()
    lyric = torch.randn(B, Ll, cfg.text_hidden_dim, generator=g).cuda()
    refer = torch.randn(B, Tref, cfg.timbre_hidden_dim, generator=g).cuda()
    tmask = torch.ones(B, Lt, dtype=torch.long)
    lmask = (torch.arange(Ll)[None, :] < torch.tensor([Ll - 37 * (i % 3) for i in range(B)])[:, None]).long()
    order = torch.arange(B)
    for _ in range(2):
        enc(text, tmask, lyric, lmask, refer, order)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 5
    for _ in range(iters):
        enc(text, tmask, lyric, lmask, refer, order)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    fl = flops(cfg, B, Lt, Ll, B, Tref)
    out = {"workload": f"condition encoder, B={B}, 256 text + 512 lyric tokens, {B} x 750-frame reference clips", "gpu_ms": round(ms, 3),
           "tflop": round(fl / 1e12, 3), "gpu_tflops": round(fl / ms / 1e9, 1)}
    if not args.no_cpu:
        from oracle import cond as o_cond
        n = max(1, min(16, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n)
        c = lambda t: t[:1].cpu()  # noqa: E731
        o_cond.condition_encoder(o_cond.CondConfig(), w, c(text), tmask[:1], c(lyric), lmask[:1], c(refer), order[:1])  # warm
        t0 = time.perf_counter()
        o_cond.condition_encoder(o_cond.CondConfig(), w, c(text), tmask[:1], c(lyric), lmask[:1], c(refer), order[:1])
        cpu_s = time.perf_counter() - t0
        out.update({"cpu_oracle_s_per_item": round(cpu_s, 3), "cpu_threads": n, "cpu_s_extrapolated_batch": round(cpu_s * B, 2),
                    "speedup_vs_cpu_oracle": round(cpu_s * B / (ms * 1e-3), 1)})
    print(out)


if __name__ == "__main__":
    main()
