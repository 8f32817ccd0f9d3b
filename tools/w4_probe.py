"""fp32-store GEMM timing + check at the metric M (used for the four-wave tile experiment of DESIGN.md section 10; that kernel variant is not in the tree)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
s = torch.cuda.current_stream().cuda_stream
for (M, N, K) in [(6000, 4096, 2048), (6000, 12288, 2048), (6000, 2048, 6144)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev, dtype=torch.float32)
    f = lambda: native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 0, None, s))   # out_dtype 0 = f32?
    for _ in range(5): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    ref = A.float() @ W.float().t()
    err = float((C - ref).norm() / ref.norm())
    print(f"M={M} N={N} K={K}: {us:.1f} us {2.0*M*N*K/us*1e-6:.0f} TF/s rel err {err:.2e}")
