#!/usr/bin/env python
"""VAE decode as two half-batch decodes on two hardware queues (round 4 probe): 8 x 30 s through one NativeVae against two handles
decoding 4 songs each concurrently (streams created before anything else: see tools/dual_chain_probe.py).  The decode is not at the
power cap like the DiT's GEMMs, and it alternates MFMA-bound (k = 7) and HBM-bound (k = 1, transposed, output) convs.
Usage: python tools/dual_vae_probe.py [--batch 8] [--frames 750]"""
import argparse, ctypes, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=750)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
streams = []
for _ in range(2):
    sp = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(sp), 1) == 0
    streams.append(torch.cuda.ExternalStream(sp.value, device=dev))
import bench  # noqa: E402
from ace355.vae import NativeVae  # noqa: E402
args.tiny, args.no_vae, args.fp8 = False, False, False
dcfg, vcfg, dit, vae_a, sd, vsd = bench.build_models(args, dev)
del dit
vae_b = NativeVae(vcfg, dev)
vae_b.load_state_dict(vsd)
B, T = args.batch, args.frames
z = torch.randn(B, 64, T, generator=torch.Generator().manual_seed(1)).to(dev)

def single():
    return vae_a.decode(z)

def dual():
    outs = [None, None]
    def work(i, v):
        with torch.cuda.stream(streams[i]):
            sl = slice(i * B // 2, (i + 1) * B // 2)
            outs[i] = v.decode(z[sl].contiguous())
            streams[i].synchronize()
    th = [threading.Thread(target=work, args=(i, v)) for i, v in enumerate((vae_a, vae_b))]
    for t in th: t.start()
    for t in th: t.join()
    return torch.cat(outs, 0)

def single_on_stream():
    with torch.cuda.stream(streams[0]):
        o = vae_a.decode(z)
        streams[0].synchronize()
    return o

for name, fn in (("single (null stream)", single), ("single (own stream)", single_on_stream), ("dual", dual)) * 2:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters): o = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / args.iters * 1e3:.2f} ms per {B} x {T}-frame decode", flush=True)
a = single(); torch.cuda.synchronize(); b = dual()
print("single == dual bit for bit:", torch.equal(a, b))
