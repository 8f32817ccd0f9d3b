#!/usr/bin/env python
"""Measurement only (never the product): the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) at the DiT's projection
shapes beside this repo's hand-written kernel, same random data, interleaved rounds.  Answers one question: is ~1000 TFLOP/s at
these shapes a property of the chip under this data (power-limited clock) or of the kernel?
Usage: python tools/lib_gemm_reference.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355  # noqa: E402
from ace355 import native  # noqa: E402
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
M = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
s = torch.cuda.current_stream().cuda_stream
for name, N, K in (("qkv", 4096, 2048), ("o_proj", 2048, 2048), ("gate_up", 12288, 2048), ("down", 2048, 6144)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    Wt = W.t()
    def ours(): native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 1, None, s))
    def vendor(): torch.matmul(A, Wt, out=C)
    res = {}
    for rnd in range(3):
        for tag, f in (("ours", ours), ("vendor", vendor)):
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(tag, []).append(e0.elapsed_time(e1) / 20 * 1e-3)
    fl = 2.0 * M * N * K
    print(f"{name:8s} M={M} N={N:5d} K={K:4d}: hand-written {fl/min(res['ours'])/1e12:7.1f} TF/s ({min(res['ours'])*1e6:6.1f} us)   "
          f"vendor library {fl/min(res['vendor'])/1e12:7.1f} TF/s ({min(res['vendor'])*1e6:6.1f} us)", flush=True)
