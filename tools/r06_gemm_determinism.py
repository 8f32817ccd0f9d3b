"""Round 6, verdict item 1: which GEMM launch of the full-size condition encoder is not reproducible?  Every (mode, M, N, K) the
lyric / timbre / text encoders launch (csrc/cond.hip: encoder_layers) through the C-ABI test hooks, REPS times on the same operands:
distinct output hashes, error vs fp32 torch.  Knobs from the environment (ACE355_*).
"""
import hashlib
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = int(os.environ.get("REPS", "12"))


def main():
    import ace355  # noqa: F401
    from ace355 import native
    lib = native.lib()
    dev = torch.device("cuda:0")
    p = native.ptr
    bf = lambda x: x.to(torch.bfloat16)  # noqa: E731
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))  # noqa: E731
    g = torch.Generator().manual_seed(1)
    shapes = []
    for M in (400, 288, 48, 750):
        shapes += [("store_f32", M, 2048, 1024), ("headnorm_rope", M, 4096, 2048), ("residual", M, 2048, 2048), ("swiglu", M, 12288, 2048),
                   ("residual", M, 2048, 6144)]
    shapes += [("store_f32", 288, 2048, 64), ("swiglu", 1000, 12288, 2048), ("swiglu", 300, 1536, 256)]
    if os.environ.get("SHAPES") == "wide":   # every mode on the launch shape of the encoder's gate|up projection, and that projection at other K / M / N
        shapes = [("store_f32", 400, 12288, 2048), ("store_bf16", 400, 12288, 2048), ("residual", 400, 12288, 2048), ("headnorm_rope", 400, 12288, 2048),
                  ("swiglu", 400, 12288, 128), ("swiglu", 400, 12288, 256), ("swiglu", 400, 12288, 512), ("swiglu", 400, 12288, 1024),
                  ("swiglu", 128, 12288, 2048), ("swiglu", 256, 12288, 2048), ("swiglu", 400, 6144, 2048), ("swiglu", 400, 24576, 2048),
                  ("swiglu", 400, 12288, 2048)]
    for kind, M, N, K in shapes:
        A = bf(torch.randn(M, K, generator=g)).to(dev)
        W = bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
        hashes, errs = [], []
        for rep in range(REPS):
            if kind == "store_f32":
                out = torch.empty(M, N, device=dev)
                native.check(lib.ace355_gemm_bf16(p(A), p(W), p(out), M, N, K, 0, None, None), kind)
                ref = A.float() @ W.float().t()
            elif kind == "store_bf16":
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                native.check(lib.ace355_gemm_bf16(p(A), p(W), p(out), M, N, K, 1, None, None), kind)
                ref = A.float() @ W.float().t()
            elif kind == "swiglu":
                out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
                native.check(lib.ace355_gemm_bf16_fused(p(A), p(W), p(out), M, N, K, 1, None, None, 0, 0, None), kind)
                Wv = W.view(N // 64, 2, 32, K)
                ref = F.silu(A.float() @ Wv[:, 0].reshape(N // 2, K).float().t()) * (A.float() @ Wv[:, 1].reshape(N // 2, K).float().t())
            elif kind == "residual":
                out = torch.ones(M, N, device=dev)
                native.check(lib.ace355_gemm_bf16_residual(p(A), p(W), p(out), M, N, K, None, None, 0, M, None, 0, None), kind)
                ref = 1.0 + A.float() @ W.float().t()
            else:
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                wq = torch.ones(128, device=dev)
                native.check(lib.ace355_gemm_bf16_headnorm(p(A), p(W), p(out), M, N, K, N // 2, N // 4 * 3, p(wq), p(wq), 1e-6, 1, M, 1e6, None), kind)
                ref = None
            torch.cuda.synchronize()
            hashes.append(hashlib.sha256(out.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:12])
            if ref is not None:
                errs.append(rel(out, ref))
        print(f"{kind:14s} M={M:5d} N={N:6d} K={K:5d}: {len(set(hashes))} distinct of {REPS}"
              + (f", rel vs fp32 min {min(errs):.2e} max {max(errs):.2e}" if errs else ""), flush=True)


if __name__ == "__main__":
    main()
