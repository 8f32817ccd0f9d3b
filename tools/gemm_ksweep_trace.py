#!/usr/bin/env python
"""Launch one GEMM per (M,N,K) triple, 6 times each in shape order; run under rocprofv3 --kernel-trace and summarise with
tools/gemm_ksweep_summary.py: true GPU-side durations per shape (the Python-side launch rate hides short kernels)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
s = torch.cuda.current_stream().cuda_stream
for (M, N, K) in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    torch.cuda.synchronize()
    for _ in range(6):
        native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 1, None, s))
    torch.cuda.synchronize()
