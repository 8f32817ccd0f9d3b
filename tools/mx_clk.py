#!/usr/bin/env python
"""ACE355_GEMM_CLK=1: in-kernel shader-clock breakdown of the MXFP8 GEMM (test hook ace355_gemm_mxfp8) at the metric shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
s = torch.cuda.current_stream().cuda_stream
M = 6000
for name, N, K, mode in [("qkv store", 4096, 2048, 0), ("gate_up swiglu", 12288, 2048, 3), ("down resid", 2048, 6144, 2)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    print(name, file=sys.stderr, flush=True)
    for _ in range(4):
        if mode == 2:
            C = torch.zeros(M, N, device=dev)
        elif mode == 3:
            C = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
        else:
            C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        native.check(lib.ace355_gemm_mxfp8(P(A), P(W), P(C), M, N, K, mode, None, None, 0, 375, s))
