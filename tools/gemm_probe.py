#!/usr/bin/env python
"""GEMM shape probe: TF/s for arbitrary M,N,K triples (bf16 store epilogue), interleaved rounds, random data.
Usage: python tools/gemm_probe.py M,N,K [M,N,K ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355  # noqa: E402
from ace355 import native  # noqa: E402

lib = native.lib()
dev = torch.device("cuda:0")
P = native.ptr
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
cases = []
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    cases.append((M, N, K, A, W, C))
s = torch.cuda.current_stream().cuda_stream
best = {}
for rnd in range(5):
    for (M, N, K, A, W, C) in cases:
        for _ in range(2):
            native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 1, None, s))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 1, None, s))
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        best.setdefault((M, N, K), []).append(t)
for (M, N, K), ts in best.items():
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"M={M:6d} N={N:6d} K={K:5d}: median {med*1e6:8.1f} us {2.0*M*N*K/med/1e12:7.1f} TF/s   best {2.0*M*N*K/ts[0]/1e12:7.1f} TF/s", flush=True)
