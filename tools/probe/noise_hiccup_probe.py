"""Where does the occasional 30-140 ms host stall inside prepare_noise + upload come from?  Times the sub-steps over many iterations with
GPU work between them (as in a bench pass) and prints the worst cases.  python tools/probe/noise_hiccup_probe.py [iters]"""
import os, sys, time
import torch

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads(), "cpus", len(os.sched_getaffinity(0)),
          "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
    try:
        print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
    except Exception as e:
        print("cpu.max n/a", e)
    rows = []
    pinned = torch.empty(8, 750, 64, dtype=torch.float32).pin_memory()
    for it in range(n):
        for _ in range(12):
            b = a @ a          # ~100 ms of GPU work
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = []
        for s in range(8):
            g = torch.Generator(device="cpu").manual_seed(1000 + s)
            outs.append(torch.randn(1, 750, 64, generator=g, dtype=torch.float32))
        t1 = time.perf_counter()
        x = torch.cat(outs, dim=0)
        t2 = time.perf_counter()
        y = x.to(dev, non_blocking=True)
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        pinned.copy_(x)
        z = pinned.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    names = ["randn x8", "cat", "to(device) pageable", "sync", "pinned copy + to + sync"]
    for i, nm in enumerate(names):
        col = sorted(r[i] for r in rows)
        print(f"{nm:28s} median {1e3 * col[len(col) // 2]:7.3f} ms  p99 {1e3 * col[int(len(col) * 0.99)]:7.3f}  max {1e3 * col[-1]:7.3f}")

main()
