#!/usr/bin/env python
"""Micro-benchmarks of the hot kernels at the DiT/VAE shapes of the metric config (runs on the GPU box).

Within-process interleaved timing (cdna guide 5.4 rule 24), random bf16 data (rule 25).  Prints TF/s per shape.
Usage: python tools/bench_kernels.py [gemm] [attn] [conv]   (env ACE355_GEMM=v1 selects the register-staged GEMM)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355  # noqa: E402
from ace355 import native  # noqa: E402

lib = native.lib()
dev = torch.device("cuda:0")
P = native.ptr


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s = torch.cuda.current_stream().cuda_stream
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def stream():
    return torch.cuda.current_stream().cuda_stream


def bench_gemm():
    M = 6000
    shapes = [("qkv", M, 4096, 2048, "store"), ("o/q_c (N=2048)", M, 2048, 2048, "store"), ("o_proj resid", M, 2048, 2048, "resid"),
              ("gate_up swiglu", M, 12288, 2048, "swiglu"), ("down resid", M, 2048, 6144, "resid"), ("patchify", M, 2048, 384, "f32"),
              ("cross q (M=3000)", 3000, 2048, 2048, "store"), ("cross o resid (M=3000)", 3000, 2048, 2048, "resid"),
              ("M=750 qkv", 750, 4096, 2048, "store"), ("M=750 down", 750, 2048, 6144, "resid"), ("M=24000 gate_up", 24000, 12288, 2048, "swiglu")]
    for name, M_, N, K, mode in shapes:
        A = torch.randn(M_, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        if mode == "store":
            C = torch.empty(M_, N, device=dev, dtype=torch.bfloat16)
            fn = lambda: native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M_, N, K, 1, None, stream()))
        elif mode == "f32":
            C = torch.empty(M_, N, device=dev, dtype=torch.float32)
            fn = lambda: native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M_, N, K, 0, None, stream()))
        elif mode == "resid":
            C = torch.zeros(M_, N, device=dev, dtype=torch.float32)
            g1 = torch.randn(N, device=dev)
            g2 = torch.randn(64, N, device=dev)
            fn = lambda: native.check(lib.ace355_gemm_bf16_fused(P(A), P(W), P(C), M_, N, K, 0, P(g1), P(g2), N, 375, stream()))
        else:
            C = torch.empty(M_, N // 2, device=dev, dtype=torch.bfloat16)
            fn = lambda: native.check(lib.ace355_gemm_bf16_fused(P(A), P(W), P(C), M_, N, K, 1, None, None, 0, 0, stream()))
        t = timeit(fn)
        print(f"gemm {name:22s} M={M_:6d} N={N:6d} K={K:5d}: {t*1e6:8.1f} us  {2.0*M_*N*K/t/1e12:7.1f} TF/s", flush=True)


def bench_attn():
    for name, N, Sq, Skv, win in [("self full", 16, 375, 375, -1), ("self band", 16, 375, 375, 128), ("cross", 16, 375, 769, -1),
                                  ("cross (cond half)", 8, 375, 769, -1), ("self full 120s", 16, 1500, 1500, -1), ("self band 120s", 16, 1500, 1500, 128)]:
        q = torch.randn(N, Sq, 2048, device=dev).to(torch.bfloat16)
        k = torch.randn(N, Skv, 1024, device=dev).to(torch.bfloat16)
        v = torch.randn(N, Skv, 1024, device=dev).to(torch.bfloat16)
        o = torch.empty_like(q)
        fn = lambda: native.check(lib.ace355_attention(P(q), P(k), P(v), P(o), N, Sq, Skv, 16, 8, win, 128 ** -0.5, stream()))
        t = timeit(fn, iters=10)
        keys = Skv if win < 0 else sum(min(Sq - 1, i + win) - max(0, i - win) + 1 for i in range(Sq)) / Sq
        fl = 4.0 * N * 16 * Sq * keys * 128
        print(f"attn {name:20s} N={N} Sq={Sq} Skv={Skv} win={win}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s (incl. V transpose + sync)", flush=True)


def bench_conv():
    B = 8
    for name, L, C, taps, dil in [("stage5 k7 d1", 1440000 // 4, 128, 7, 1), ("stage5 k7 d9", 1440000 // 4, 128, 7, 9), ("stage5 k1", 1440000 // 4, 128, 1, 1),
                                  ("stage3 k7 d3", 180000, 256, 7, 3), ("stage1 k7 d1", 7500, 1024, 7, 1)]:
        x = torch.randn(B, L, C, device=dev).to(torch.bfloat16)
        w = (torch.randn(C, taps, C, device=dev) / (C * taps) ** 0.5).to(torch.bfloat16)
        b = torch.zeros(C, device=dev)
        al = torch.zeros(C, device=dev)
        y = torch.empty_like(x)
        fn = lambda: native.check(lib.ace355_conv1d_nlc(P(x), P(w), P(b), P(al), P(al), None, P(y), B, L, C, C, taps, dil, stream()))
        t = timeit(fn, iters=5, warm=2)
        fl = 2.0 * B * L * C * C * taps
        print(f"conv {name:16s} B={B} L={L} C={C}: {t*1e6:9.1f} us  {fl/t/1e12:7.1f} TF/s  {(2*B*L*C*2)/t/1e9:7.0f} GB/s(min traffic)", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "attn", "conv"]
    print("variant:", os.environ.get("ACE355_GEMM", "v2 (glds)"))
    if "gemm" in which:
        bench_gemm()
    if "attn" in which:
        bench_attn()
    if "conv" in which:
        bench_conv()
