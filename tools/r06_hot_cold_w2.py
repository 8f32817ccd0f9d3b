#!/usr/bin/env python
"""Round 6, follow-up of r06_hot_cold_w.py: WHICH cache makes a small GEMM faster when its weights are warm?  Per launch, timed with events
around the GEMM alone: (1) cold weights (rotation beyond the MALL), (2) cold weights READ by another kernel right before (a torch reduction
over the buffer: through whatever XCD's L2 its workgroups sit on, and through the MALL), (3) the same after reading 64 MB / 200 MB of other
data in between (does it survive in the MALL?), (4) hot (the same buffer as the launch before)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355  # noqa: E402,F401
from ace355 import native  # noqa: E402

lib = native.lib()
dev = torch.device("cuda:0")
P = native.ptr


def stream():
    return torch.cuda.current_stream().cuda_stream


def run(name, M, N, K, mode, iters=120):
    nb = max(2, int(640e6 // (N * K * 2)) + 1)
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(nb)]
    junk = torch.randn(50_000_000, device=dev)   # 200 MB
    if mode == "store":
        C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        call = lambda W: native.check(lib.ace355_gemm_bf16(P(A), P(W), P(C), M, N, K, 1, None, stream()))
    else:
        C = torch.zeros(M, N, device=dev, dtype=torch.float32)
        g1, g2 = torch.randn(N, device=dev), torch.randn(64, N, device=dev)
        call = lambda W: native.check(lib.ace355_gemm_bf16_fused(P(A), P(W), P(C), M, N, K, 0, P(g1), P(g2), N, 375, stream()))

    def timed(pre):
        tot = 0.0
        evs = []
        for i in range(iters):
            W = Ws[i % nb]
            pre(W)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call(W)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        return ts[len(ts) // 2]
    res = {}
    res["cold"] = timed(lambda W: None)
    res["read before"] = timed(lambda W: W.view(torch.int16).amax())
    res["read, then 64 MB of other reads"] = timed(lambda W: (W.view(torch.int16).amax(), junk[:16_000_000].amax()))
    res["read, then 200 MB of other reads"] = timed(lambda W: (W.view(torch.int16).amax(), junk.amax()))
    res["launched twice (hot)"] = timed(lambda W: call(W))
    print(f"{name:24s} M={M:4d} N={N:5d} K={K:4d} (W {N * K * 2 / 1e6:.0f} MB): " + ", ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    run("cross-q-like (store)", 375, 2048, 2048, "store")
    run("QKV-like (store)", 750, 4096, 2048, "store")
    run("down (residual)", 750, 2048, 6144, "resid")
    run("gate|up-sized (store)", 750, 12288, 2048, "store")
