#!/usr/bin/env python
"""Small-M (batch-1 request: M = 750 / 375 rows) GEMM timings per projection, HIP events over 40 launches each.
Env knobs of gemm.hip apply (ACE355_GEMM_DEEP / KSPLIT / BIG / CLK).  Usage: small_m_probe.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
s = torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 750
clk = os.environ.get("ACE355_GEMM_CLK", "0") != "0"
shapes = [("qkv(store)", M, 4096, 2048, "store"), ("o_proj", M, 2048, 2048, "resid"), ("cross_q(store)", M // 2, 2048, 2048, "store"),
          ("cross_o", M // 2, 2048, 2048, "resid"), ("gate_up", M, 12288, 2048, "swiglu"), ("down", M, 2048, 6144, "resid")]
tot = 0.0
for name, m, N, K, mode in shapes:
    A = torch.randn(m, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    if mode == "store":
        C = torch.empty(m, N, device=dev, dtype=torch.bfloat16)
        f = lambda: native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), m, N, K, 1, None, s))
    elif mode == "resid":
        C = torch.zeros(m, N, device=dev); g1 = torch.randn(N, device=dev); g2 = torch.randn(64, N, device=dev)
        f = lambda: native.check(lib.ace355_gemm_bf16_residual(P(A), P(W), P(C), m, N, K, P(g1), P(g2), N, 375, None, 0, s))
    else:
        C = torch.empty(m, N // 2, device=dev, dtype=torch.bfloat16)
        f = lambda: native.check(lib.ace355_gemm_bf16_fused(P(A), P(W), P(C), m, N, K, 1, None, None, 0, 0, s))
    reps = 3 if clk else 40
    if clk: print(name, file=sys.stderr, flush=True)
    for _ in range(3): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tot += us
    if not clk: print(f"{name:16s} M={m:4d} N={N:5d} K={K:4d}: {us:7.1f} us  {2.0*m*N*K/us*1e-6:7.1f} TF/s")
if not clk: print(f"sum {tot:.1f} us")
