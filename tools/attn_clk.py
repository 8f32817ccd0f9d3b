#!/usr/bin/env python
"""Run with ACE355_ATTN_CLK=1: prints the in-kernel shader-clock breakdown (prologue, K loop, wait+barrier) of the attention kernel
at the metric shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ace355
from ace355 import native
lib = native.lib(); dev = torch.device("cuda:0"); P = native.ptr
for (N, Sq, Skv, win) in [(16, 375, 375, -1), (16, 375, 375, 128), (8, 375, 769, -1)]:
    q = torch.randn(N, Sq, 2048, device=dev).to(torch.bfloat16); k = torch.randn(N, Skv, 1024, device=dev).to(torch.bfloat16)
    v = torch.randn(N, Skv, 1024, device=dev).to(torch.bfloat16); o = torch.empty_like(q)
    for _ in range(6):
        native.check(lib.ace355_attention(P(q), P(k), P(v), P(o), N, Sq, Skv, 16, 8, win, 128 ** -0.5, None))
