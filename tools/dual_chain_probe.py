#!/usr/bin/env python
"""Two half-batch samplers side by side (round 4).  One sampler at B songs against two independent samplers at B/2 songs each on two
streams (two handles, two host threads), the second started `--offset-ms` later so that the two launch sequences are out of phase.
Run it twice: as is (every launch shaped for 256 CUs: the two chains' workgroups interleave over the whole chip) and with
ACE355_MAX_WGS=128 (every launch shaped for half the chip: the chains run side by side on 128 CUs each, and one chain's memory
bursts - residual read-modify-writes, attention prologues - fall into the other's MFMA phases).
Usage: [ACE355_MAX_WGS=128] python tools/dual_chain_probe.py [--batch 8] [--steps 27] [--offset-ms 0.3]"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ace355.dit import SLOT_COND, SLOT_NULL, NativeDit, schedule  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=27)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--offset-ms", type=float, nargs="*", default=[0.0, 0.35])
    ap.add_argument("--skip-single", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.init()
    torch.zeros(1, device=dev)
    # Two HIP streams created back to back BEFORE anything else creates a stream (the handles' side streams, torch's pool): the runtime
    # hands out at most GPU_MAX_HW_QUEUES (4) hardware queues and shares them beyond that; the first runs of this probe had both chains on
    # ONE queue, strictly serialised (rocprofv3 Queue_Id 4 for every launch) - torch.cuda.Stream() pool streams and late-created
    # streams alike - which also explains why the two-stream probes of rounds 2 / 3 measured "no difference".
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    streams = []
    for _ in range(2):
        sp = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(sp), 1) == 0   # hipStreamNonBlocking
        streams.append(torch.cuda.ExternalStream(sp.value, device=dev))
    args.tiny, args.no_vae, args.fp8 = False, True, False
    dcfg, _, dit_a, _, sd, _ = bench.build_models(args, dev)
    dit_b = NativeDit(dcfg, dev)
    dit_b.load_state_dict(sd)
    B, T, L = args.batch, 750, 769
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(L, dcfg.hidden_size, generator=g).to(dev)
    null = torch.randn(1, dcfg.hidden_size, generator=g).to(dev)
    ctx = torch.randn(B, T, 128, generator=g).to(dev)
    noise = torch.randn(B, T, 64, generator=g).to(dev)
    ts = schedule(args.steps, 1.0, None)
    for d in (dit_a, dit_b):
        d.set_condition(SLOT_COND, enc)
        d.set_condition(SLOT_NULL, null.reshape(1, -1), L=L)

    def run_single():
        return dit_a.sample(noise, ctx, ts, guidance_scale=7.0)

    def run_dual(offset_ms):
        outs = [None, None]

        def work(i, d):
            if i == 1 and offset_ms > 0:
                time.sleep(offset_ms * 1e-3)
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

    cases = [] if args.skip_single else [("single", run_single)]
    cases += [(f"dual, second chain {o:g} ms later", (lambda o=o: run_dual(o))) for o in args.offset_ms]
    print(f"ACE355_MAX_WGS={os.environ.get('ACE355_MAX_WGS', '256 (default)')}", flush=True)
    for name, fn in cases * 2:
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            out = fn()
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / args.iters * 1e3:.1f} ms per {B}-song sampler pass", flush=True)
    if not args.skip_single:
        a = run_single()
        torch.cuda.synchronize()
        b = run_dual(0.0)
        print("max |single - dual| =", float((a - b).abs().max()))


if __name__ == "__main__":
    main()
