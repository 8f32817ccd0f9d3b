#!/usr/bin/env python
"""Attention at the BATCH-1 request's shapes (2 x 375 self, 1 x 375 over 769 keys split-KV) through the unit hook; with
ACE355_ATTN_CLK=1 the kernel's own clock probe prints prologue / loop / whole-wave cycles; under rocprofv3 the launch durations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
for (N, Sq, Skv, win) in [(2, 375, 375, -1), (2, 375, 375, 128), (1, 375, 769, -1), (4, 375, 375, -1), (2, 375, 769, -1)]:
    g = torch.Generator(device=dev).manual_seed(N + Sq + Skv)
    q = torch.randn(N, Sq, 2048, device=dev, generator=g).to(torch.bfloat16); k = torch.randn(N, Skv, 1024, device=dev, generator=g).to(torch.bfloat16)
    v = torch.randn(N, Skv, 1024, device=dev, generator=g).to(torch.bfloat16); o = torch.empty_like(q)
    for _ in range(6):
        native.check(lib.ace355_attention(P(q), P(k), P(v), P(o), N, Sq, Skv, 16, 8, win, 128 ** -0.5, None))
    torch.cuda.synchronize()
    print(f"N={N} Sq={Sq} Skv={Skv} win={win} done", flush=True)
