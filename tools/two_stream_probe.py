#!/usr/bin/env python
"""Potential of running the two CFG halves as two concurrent streams: one sampler at B songs (2B sequences per forward) vs two
independent samplers at B/2 songs each on two HIP streams from two host threads.  Same total work; the difference is what
kernel-level overlap (one half's epilogues / elementwise kernels under the other half's MFMA loops) can buy.
Usage: python tools/two_stream_probe.py [--batch 8] [--steps 27]"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ace355.dit import SLOT_COND, SLOT_NULL, schedule  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=27)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    args.tiny, args.no_vae = False, True
    dcfg, _, dit_a, _, sd, _ = bench.build_models(args, dev)
    from ace355.dit import NativeDit
    dit_b = NativeDit(dcfg, dev)
    dit_b.load_state_dict(sd)
    B, T, L = args.batch, 750, 769
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(L, dcfg.hidden_size, generator=g).to(dev)
    null = torch.randn(1, dcfg.hidden_size, generator=g).to(dev)
    ctx = torch.randn(B, T, 128, generator=g).to(dev)
    noise = torch.randn(B, T, 64, generator=g).to(dev)
    ts = schedule(args.steps, 3.0, None)
    for d in (dit_a, dit_b):
        d.set_condition(SLOT_COND, enc)
        d.set_condition(SLOT_NULL, null.reshape(1, -1), L=L)

    def run_single():
        return dit_a.sample(noise, ctx, ts, guidance_scale=7.0)

    def run_dual():
        outs = [None, None]
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]

        def work(i, d):
            with torch.cuda.stream(streams[i]):
                sl = slice(i * B // 2, (i + 1) * B // 2)
                outs[i] = d.sample(noise[sl], ctx[sl], ts, guidance_scale=7.0)
                streams[i].synchronize()
        th = [threading.Thread(target=work, args=(i, d)) for i, d in enumerate((dit_a, dit_b))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return torch.cat(outs, 0)

    for name, fn in (("single", run_single), ("dual", run_dual), ("single", run_single), ("dual", run_dual)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / args.iters * 1e3:.1f} ms per {B}-song sampler pass", flush=True)
    a = run_single()
    torch.cuda.synchronize()  # the same handle must not run on two streams at once
    b = run_dual()
    print("max |single - dual| =", float((a - b).abs().max()))


if __name__ == "__main__":
    main()
