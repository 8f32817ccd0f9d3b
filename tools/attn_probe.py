#!/usr/bin/env python
"""Attention at the metric's shapes through the unit hook: error against fp32 torch SDPA on the same bf16 inputs, and (under
rocprofv3 --kernel-trace --stats) the kernel's own duration.  Env switches of attn.hip apply (ACE355_ATTN_1STAGE, ACE355_ATTN_GQA)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
for (N, Sq, Skv, win) in [(16, 375, 375, -1), (16, 375, 375, 128), (8, 375, 769, -1), (4, 1500, 1500, -1)]:
    g = torch.Generator(device=dev).manual_seed(N + Sq + Skv)
    q = torch.randn(N, Sq, 2048, device=dev, generator=g).to(torch.bfloat16); k = torch.randn(N, Skv, 1024, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(N, Skv, 1024, device=dev, generator=g).to(torch.bfloat16); o = torch.empty_like(q)
    for _ in range(8):
        native.check(lib.ace355_attention(P(q), P(k), P(v), P(o), N, Sq, Skv, 16, 8, win, 128 ** -0.5, None))
    torch.cuda.synchronize()
    qh = q.float().view(N, Sq, 16, 128).transpose(1, 2); kh = k.float().view(N, Skv, 8, 128).transpose(1, 2).repeat_interleave(2, 1)
    vh = v.float().view(N, Skv, 8, 128).transpose(1, 2).repeat_interleave(2, 1)
    sc = (qh @ kh.transpose(-1, -2)) * 128 ** -0.5
    if win >= 0:
        i = torch.arange(Sq, device=dev)[:, None]; j = torch.arange(Skv, device=dev)[None]
        sc = sc.masked_fill((i - j).abs() > win, float("-inf"))
    ref = (sc.softmax(-1) @ vh).transpose(1, 2).reshape(N, Sq, 2048)
    print(f"N={N} Sq={Sq} Skv={Skv} win={win}: rel L2 vs fp32 {float((o.float() - ref).norm() / ref.norm()):.3e}", flush=True)
