#!/usr/bin/env python
"""Audio tokenizer (the other direction of SURVEY 8f row N2) timing at the metric shape: B = 8 songs x 750 latent frames (30 s) ->
150 five-Hz tokens each (acoustic projection + attention pooler natively, FSQ in torch), plus parity vs the fp32 oracle on one item
and the oracle's CPU time on this box.  Usage: python tests/perf/bench_tok.py [--no-cpu]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ace355  # noqa: E402
from ace355 import weightgen  # noqa: E402
from ace355.lmhints import NativeAudioTokenizer  # noqa: E402
from oracle import detok as o_detok  # noqa: E402  (the weight-name table lives with the oracle; the timing below is the product's)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    cfg = ace355.DetokConfig()
    o_cfg = o_detok.DetokConfig()
    w = weightgen.make_dit_weights(o_detok.tok_weight_shapes(o_cfg), cfg.hidden_size, seed=3, mode="init")
    g = torch.Generator().manual_seed(1)
    q = {"quantizer.project_in.weight": 0.05 * torch.randn(6, cfg.hidden_size, generator=g), "quantizer.project_in.bias": torch.zeros(6),
         "quantizer.project_out.weight": 0.5 * torch.randn(cfg.hidden_size, 6, generator=g), "quantizer.project_out.bias": torch.zeros(cfg.hidden_size)}
    tok = NativeAudioTokenizer(cfg, "cuda:0")
    tok.load_state_dict({**w, **q})
    B, T = 8, 750
    x = torch.randn(B, T, 64, generator=g).cuda()
    for _ in range(2):
        quant, idx = tok.tokenize(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        quant, idx = tok.tokenize(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    D, Fh, P = cfg.hidden_size, cfg.intermediate_size, cfg.pool_window_size
    qd, kv = cfg.num_attention_heads * cfg.head_dim, cfg.num_key_value_heads * cfg.head_dim
    frames, rows = B * T, B * (T // P) * (P + 1)
    fl = 2 * frames * (64 * D + D * D) + rows * cfg.num_attention_pooler_hidden_layers * 2 * (D * (qd + 2 * kv) + qd * D + 3 * D * Fh)
    out = {"workload": f"audio tokenizer, B={B} x {T} frames -> {B * (T // P)} tokens", "gpu_ms": round(ms, 3), "tflop": round(fl / 1e12, 3),
           "gpu_tflops": round(fl / ms / 1e9, 1)}
    if not args.no_cpu:
        n = max(1, min(16, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n)
        xs = x[:1].cpu().reshape(1, T // P, P, 64)
        o_detok.tokenizer_pool(o_cfg, w, xs[:, :10])
        t0 = time.perf_counter()
        ref = o_detok.tokenizer_pool(o_cfg, w, xs)
        cpu_s = time.perf_counter() - t0
        rel = float((tok.pool(x[:1]).cpu() - ref).norm() / ref.norm())
        out.update({"rel_l2_vs_oracle": round(rel, 5), "cpu_oracle_s_per_item": round(cpu_s, 3), "cpu_threads": n,
                    "speedup_vs_cpu_oracle": round(cpu_s * B / (ms * 1e-3), 1)})
    print(out)


if __name__ == "__main__":
    main()
