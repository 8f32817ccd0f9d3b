#!/usr/bin/env python
"""Run with ACE355_GEMM_CLK=1: prints the in-kernel shader-clock breakdown (K loop, prologue, epilogue) per GEMM mode."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
s = torch.cuda.current_stream().cuda_stream
M = int(os.environ.get("GEMM_CLK_M", "6000"))
for name, N, K, mode in [("qkv store", 4096, 2048, "store"), ("o_proj resid", 2048, 2048, "resid"), ("gate_up swiglu", 12288, 2048, "swiglu"), ("down resid", 2048, 6144, "resid")]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    print(name, file=sys.stderr, flush=True)
    for _ in range(12):
        if mode == "store":
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 1, None, s))
        elif mode == "resid":
            C = torch.zeros(M, N, device=dev); g1 = torch.randn(N, device=dev); g2 = torch.randn(64, N, device=dev)
            native.check(lib.ace355_gemm_bf16_fused(P(A), P(W), P(C), M, N, K, 0, P(g1), P(g2), N, 375, s))
        else:
            C = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
            native.check(lib.ace355_gemm_bf16_fused(P(A), P(W), P(C), M, N, K, 1, None, None, 0, 0, s))
