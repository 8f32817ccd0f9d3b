#!/usr/bin/env python
"""Detokenizer timing at the metric shape: B=8 songs x 150 five-Hz tokens (30 s) -> [8, 750, 64] hints,
plus parity vs the fp32 oracle on a bounded sample and the oracle's CPU time on this box.
Usage: python tests/perf/bench_detok.py [--no-cpu]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ace355  # noqa: E402
from ace355 import weightgen  # noqa: E402
from ace355.lmhints import NativeDetokenizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    cfg = ace355.DetokConfig()
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=3, mode="init")
    det = NativeDetokenizer(cfg, "cuda:0")
    det.load_state_dict(w)
    B, T5 = 8, 150
    x = torch.randn(B, T5, cfg.hidden_size, generator=torch.Generator().manual_seed(0)).cuda()
    for _ in range(2):
        y = det(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = det(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    D, Fh = cfg.hidden_size, cfg.intermediate_size
    q, kv = cfg.num_attention_heads * cfg.head_dim, cfg.num_key_value_heads * cfg.head_dim
    rows = B * T5 * cfg.pool_window_size
    fl = 2 * B * T5 * D * D + rows * (cfg.num_attention_pooler_hidden_layers * 2 * (D * (q + 2 * kv) + q * D + 3 * D * Fh) + 2 * D * 64)
    out = {"workload": f"detokenizer, B={B} x {T5} tokens -> {rows} frames", "gpu_ms": round(ms, 3), "tflop": round(fl / 1e12, 3),
           "gpu_tflops": round(fl / ms / 1e9, 1)}
    if not args.no_cpu:
        from oracle import detok as o_detok
        n = max(1, min(16, len(os.sched_getaffinity(0))))
        torch.set_num_threads(n)
        xs = x[:1].cpu()
        o_detok.detokenizer(o_detok.DetokConfig(), w, xs[:, :10])
        t0 = time.perf_counter()
        ref = o_detok.detokenizer(o_detok.DetokConfig(), w, xs)
        cpu_s = time.perf_counter() - t0
        rel = float((y[:1].cpu() - ref).norm() / ref.norm())
        out.update({"rel_l2_vs_oracle": round(rel, 5), "cpu_oracle_s_per_item": round(cpu_s, 3), "cpu_threads": n,
                    "speedup_vs_cpu_oracle": round(cpu_s * B / (ms * 1e-3), 1)})
    print(out)


if __name__ == "__main__":
    main()
