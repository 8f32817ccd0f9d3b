#!/usr/bin/env python
"""GEMM time vs K at fixed M,N: the intercept is the per-kernel fixed cost (launch + prologue + epilogue + tail)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
st = lambda: torch.cuda.current_stream().cuda_stream
for (M, N) in [(6000, 2048), (6000, 4096), (3000, 2048)]:
    for mode in ("store", "resid", "f32"):
        row = []
        for K in (64, 128, 256, 512, 1024, 2048, 4096):
            A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
            if mode == "store":
                C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                fn = lambda: native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 1, None, st()))
            elif mode == "f32":
                C = torch.empty(M, N, device=dev, dtype=torch.float32)
                fn = lambda: native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 0, None, st()))
            else:
                C = torch.zeros(M, N, device=dev); g1 = torch.randn(N, device=dev); g2 = torch.randn(64, N, device=dev)
                fn = lambda: native.check(lib.ace355_gemm_bf16_fused(P(A), P(W), P(C), M, N, K, 0, P(g1), P(g2), N, 375, st()))
            row.append("%6.1f" % timeit(fn))
        print(f"M={M} N={N} {mode:6s} K=64..4096 us:", " ".join(row), flush=True)
# empty-ish kernel launch cost for reference
x = torch.zeros(1024, device=dev)
print("torch tiny kernel us: %.1f" % timeit(lambda: x.add_(1.0)))
