#!/usr/bin/env python
"""bench.py - songs/s + RTF of the native denoise + decode path (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input on every rank:
  RCCL broadcast of the conditioning bundle (N > 1) -> cross-K/V build for the cond / null slots ->
  27-step flow-matching sampler with CFG 7.0 + APG (2B sequences per DiT forward) -> Oobleck decode to
  48 kHz stereo fp32 -> peak normalise.   Workload at N=1: 30 s audio, 27 steps, batch 8 (the metric's config).
Multi-GPU: `value` at N > 1 is the STRONG split of the metric's ONE batch of `--batch` (8) songs over the ranks in
contiguous slices - 8/4/2/1 songs per rank at 1/2/4/8 GPUs - through `ace355.dist.run_request` (exact-size RCCL broadcast of the
request -> shard -> per-rank sampler + decode); the weak number (the reference's per-call cap of 8 songs on EVERY rank,
handler/service_generate_request.py:12) is measured in the same run and reported beside it under `weak`.  `--scaling weak` makes the
weak number the headline instead.  `python bench.py --gpus N` with N > 1 outside torchrun re-executes itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU over RCCL).

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job songs/s with inputs resident
in HBM, plus `roofline` (dominant kernel = the bf16 MFMA GEMM, HIP-event timed) and `cpu_baseline`
(the oracle/ restatement timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_MXFP8_TFLOPS = 5000.0  # dense MX-scaled fp8 MFMA (same guide; measured ceiling there 4647-4686 TFLOP/s)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--duration", type=float, default=30.0, help="seconds of audio per song")
    ap.add_argument("--infer-steps", type=int, default=27)
    ap.add_argument("--batch", type=int, default=8, help="songs per request (strong: in total; weak: per rank)")
    ap.add_argument("--enc-len", type=int, default=769, help="encoder tokens (256 text + 512 lyric + 1 timbre)")
    ap.add_argument("--guidance", type=float, default=7.0)
    ap.add_argument("--no-vae", action="store_true", help="DiT-only (BASELINE configs 0/1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gc-on", action="store_true", help="leave Python's cyclic collector enabled inside the timed passes (default: collected "
                    "once, then disabled for the timed region; config.gc_disabled_in_timed_region records which)")
    ap.add_argument("--tiny", action="store_true", help="tiny architecture (smoke/debug only; result is not a benchmark)")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE configs[4] precision: the four big projections on MXFP8 MFMA (not the headline metric, whose dtype is bf16)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="strong (default, SURVEY 8e): --batch songs in total, split over the ranks, the weak number reported beside it; "
                         "weak: --batch songs on EVERY rank")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective check WITHOUT a GPU: ranks over gloo on CPU tensors run broadcast -> (scatter) -> barrier "
                         "-> MAX-reduce and rank 0 prints a line marked dry_run (not a measurement; tests/test_dist_cpu.py drives it)")
    ap.add_argument("--lm-hints", action="store_true",
                    help="per-item LM hints [G,T,64] produced on rank 0 and SCATTERED to their owners each pass (think-mode conditioning)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) started as a plain process: launch N ranks of this same command, one per GPU, over
    RCCL - the command line the driver itself uses for N > 1 - and pass its output through."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL / cross-process device memory on this driver
    raise SystemExit(subprocess.call(cmd, env=env))


def synth_weights_gpu(shapes, hidden, device, seed, kind):
    """Random-init weights of the real architecture, generated on the device (fast; same on every rank)."""
    g = torch.Generator(device=device).manual_seed(seed)
    for name, shape in shapes.items():
        if kind == "dit":
            if name.endswith("scale_shift_table"):
                w = torch.randn(shape, device=device, generator=g) / hidden ** 0.5
            elif name.endswith("norm.weight") or name.endswith("norm_out.weight"):
                w = torch.ones(shape, device=device)
            elif name.endswith(".bias"):
                w = torch.zeros(shape, device=device)
            else:
                w = 0.02 * torch.randn(shape, device=device, generator=g)
        else:
            from ace355.weightgen import _vae_gain
            if name.endswith("weight_v"):
                fan_in = shape[0] * 2 if ".conv_t1." in name else shape[1] * shape[2]
                w = torch.randn(shape, device=device, generator=g) / fan_in ** 0.5
            elif name.endswith("weight_g"):
                w = None  # filled after its weight_v
            elif name.endswith(".bias") or name.endswith(".alpha") or name.endswith(".beta"):
                w = torch.zeros(shape, device=device)
        yield name, shape, w


def build_models(args, device):
    import ace355
    from ace355.dit import NativeDit
    from ace355.vae import NativeVae
    if args.tiny:
        dcfg = ace355.DitConfig(hidden_size=256, intermediate_size=768, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1)
        vcfg = ace355.VaeConfig(decoder_channels=64, channel_multiples=(1, 2, 4), downsampling_ratios=(2, 4, 6))
    else:
        dcfg, vcfg = ace355.DitConfig(), ace355.VaeConfig()
    dit = NativeDit(dcfg, device)
    sd = {}
    for name, shape, w in synth_weights_gpu(dcfg.weight_shapes(), dcfg.hidden_size, device, 1234, "dit"):
        sd[name] = w
    dit.load_state_dict(sd)
    if getattr(args, "fp8", False):
        dit.set_precision("mxfp8")
    vae = None
    vsd = {}
    if not args.no_vae:
        vae = NativeVae(vcfg, device)
        for name, shape, w in synth_weights_gpu(vcfg.weight_shapes(), 0, device, 4321, "vae"):
            vsd[name] = w
        from ace355.weightgen import _vae_gain
        for name in list(vsd):
            if name.endswith("weight_g"):
                v = vsd[name[:-1] + "v"]
                vsd[name] = v.reshape(v.shape[0], -1).norm(dim=1).reshape(vcfg.weight_shapes()[name]) * _vae_gain(name)
        vae.load_state_dict(vsd)
    return dcfg, vcfg, dit, vae, sd, vsd


def dit_flops_per_forward_per_seq(cfg, S, L):
    """F_fwd(S, L) with band-limited sliding layers, dense full/cross attention, cached cross K/V."""
    D, Fh, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    q, kv = cfg.num_attention_heads * hd, cfg.num_key_value_heads * hd
    macs_tok = (q * D + 2 * kv * D + D * q) + (q * D + D * q) + 3 * D * Fh  # 9 projections
    W = cfg.sliding_window
    p_band = sum(min(S - 1, i + W) - max(0, i - W) + 1 for i in range(S))
    n_sl = sum(1 for t in cfg.layer_types if t == "sliding_attention")
    n_fl = cfg.num_hidden_layers - n_sl
    attn = 4 * q * (n_fl * S * S + n_sl * p_band + cfg.num_hidden_layers * S * L)
    io = 2 * S * (2 * cfg.in_channels * D + D * 2 * cfg.audio_acoustic_hidden_dim)
    return 2 * cfg.num_hidden_layers * S * macs_tok + attn + io


def vae_flops_per_frame(vcfg):
    f = 2 * vcfg.decoder_input_channels * vcfg.block_dims()[0][0] * 7
    rate = 1
    for cin, cout, s in vcfg.block_dims():
        rate *= s
        f += rate * (2 * 2 * cin * cout)          # transposed conv: 2 taps per output sample
        f += rate * 3 * (2 * cout * cout * 8)      # 3 residual units: k7 + k1
    f += rate * 2 * vcfg.decoder_channels * vcfg.audio_channels * 7
    return f


def library_source_sha() -> str:
    """sha256 (first 16 hex digits) over the sources libace355.so is built from (csrc/*.hip, csrc/common.h, include/ace355.h, sorted
    by name): computable on the GPU box, where there is no .git - tools/pmc_summary.py stamps every PMC summary with it, so a profile
    can be matched to the library that is being timed."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "ace-step-1.5-for-windows_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "*.hip")) + [os.path.join(csrc, "common.h"), os.path.join(ROOT, "include", "ace355.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_bytes_per_launch():
    """L2-miss-side bytes per GEMM launch from the committed rocprofv3 PMC passes of this same command (profiles/*.json,
    separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM: gfx950 tallies
    128-B requests at 64 B; both counters are in KiB).  These counters sit between the XCDs' L2s and the fabric: the bytes
    include Infinity-Cache hits, i.e. they bound HBM traffic from above.  bench.py cannot collect PMCs in its own process, so the
    number is only reported as `traffic` when the profile was taken on THIS library (the summary's `lib_src_sha` equals the sha of
    the sources in the tree); an older profile is named with its commit under `traffic_stale` and `traffic` stays null.
    -> dict(bytes, file, commit, lib_src_sha, current)."""
    try:
        import glob
        # the profile taken on THIS library if there is one (source sha stamped into the summary), else the last by name (reported as stale)
        # (the current round's files sit at the top of profiles/, earlier rounds' under profiles/rNN/)
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_FETCH_SIZE.json")) +
                       glob.glob(os.path.join(ROOT, "profiles", "r*", "r*_pmc_FETCH_SIZE.json")), key=os.path.basename)
        cur = library_source_sha()
        same = [c for c in cands if json.load(open(c)).get("_meta", {}).get("lib_src_sha") == cur]
        f = (same or cands)[-1]
        w = f.replace("FETCH_SIZE", "WRITE_SIZE")
        jf, jw = json.load(open(f)), json.load(open(w))
        b = (2.0 * jf["gemm"]["FETCH_SIZE"]["per_launch"] + jw["gemm"]["WRITE_SIZE"]["per_launch"]) * 1024.0
        meta = jf.get("_meta", {})
        sha = meta.get("lib_src_sha")
        return {"bytes": b, "file": os.path.relpath(f, ROOT), "commit": meta.get("commit"), "lib_src_sha": sha,
                "current": bool(sha) and sha == library_source_sha() and jw.get("_meta", {}).get("lib_src_sha") == sha}
    except Exception:
        return None


def usable_cpus() -> int:
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (a 128-thread default on a
    quota-limited container oversubscribes and slows the CPU baseline several-fold)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(args, dcfg, vcfg, sd, vsd, enc_cpu, null_cpu, ctx_cpu, T, L):
    """Time the oracle (port of the reference's CPU path) on this box's host cores on a bounded sample."""
    from oracle import dit as o_dit
    from oracle import oobleck as o_vae
    from oracle import sampler as o_sampler
    threads = usable_cpus()
    torch.set_num_threads(threads)
    w = {k: v.float().cpu() for k, v in sd.items()}
    o_cfg = o_dit.DitConfig(hidden_size=dcfg.hidden_size, intermediate_size=dcfg.intermediate_size,
                            num_hidden_layers=dcfg.num_hidden_layers, num_attention_heads=dcfg.num_attention_heads,
                            num_key_value_heads=dcfg.num_key_value_heads, sliding_window=dcfg.sliding_window)
    # DiT: batch 1 with CFG (2 sequences), first step fills the cross K/V, then 2 steady steps are timed
    enc2 = torch.cat([enc_cpu[None], null_cpu.reshape(1, 1, -1).expand(1, L, -1)], 0)
    ctx2 = ctx_cpu[:1].repeat(2, 1, 1)
    x = torch.randn(2, T, 64, generator=torch.Generator().manual_seed(1))
    cache = o_dit.CrossCache()
    tt = torch.full((2,), 0.9)
    with torch.no_grad():
        t0 = time.perf_counter()
        o_dit.dit_forward(o_cfg, w, x, tt, tt, enc2, ctx2, cache)
        t_first = time.perf_counter() - t0
        n_steady = 2
        t0 = time.perf_counter()
        for _ in range(n_steady):
            o_dit.dit_forward(o_cfg, w, x, tt, tt, enc2, ctx2, cache)
        t_step = (time.perf_counter() - t0) / n_steady
    dit_song_s = t_first + (args.infer_steps - 1) * t_step
    vae_song_s = 0.0
    vae_T = 0
    if vsd:
        vw = {k: v.float().cpu() for k, v in vsd.items()}
        ocfg = o_vae.VaeConfig(decoder_channels=vcfg.decoder_channels, channel_multiples=tuple(vcfg.channel_multiples),
                               downsampling_ratios=tuple(vcfg.downsampling_ratios))
        vae_T = 16
        z = torch.randn(1, 64, vae_T, generator=torch.Generator().manual_seed(2))
        with torch.no_grad():
            t0 = time.perf_counter()
            o_vae.decode(ocfg, vw, z)
            t_dec = time.perf_counter() - t0
        vae_song_s = t_dec * (T / vae_T)
    song_s = dit_song_s + vae_song_s
    # BASELINE.json configs[0] (the reference's CPU-runnable case: 10 s audio, 10 steps, batch 1, DiT-only, CFG 7 + APG) run IN FULL
    # through the oracle's sampler: a measured end-to-end CPU number next to the extrapolated one
    cfg0 = None
    if not args.tiny and os.environ.get("ACE355_BENCH_CFG0", "1") != "0":
        T0, steps0 = 250, 10
        ctx0 = ctx_cpu[:1, :T0].contiguous() if ctx_cpu.shape[1] >= T0 else ctx_cpu[:1].repeat(1, -(-T0 // ctx_cpu.shape[1]), 1)[:, :T0].contiguous()
        with torch.no_grad():
            t0 = time.perf_counter()
            o_sampler.generate_audio(o_cfg, w, null_cpu.reshape(1, 1, -1), enc_cpu[None], ctx0, seed=[1000], infer_steps=steps0,
                                     diffusion_guidance_sale=args.guidance)
            c0 = time.perf_counter() - t0
        cfg0 = {"config": "configs[0]: 10 s audio, 10 steps, batch 1, DiT-only, CFG on", "seconds": c0, "songs_per_s": 1.0 / c0,
                "tflops": 2 * steps0 * dit_flops_per_forward_per_seq(dcfg, (T0 + 1) // 2, L) / 1e12 / c0, "measured": "in full"}
    return {
        "config0_full_run": cfg0,
        "value": 1.0 / song_s, "unit": "songs/s", "cores": threads, "kind": "port",
        "sample": (f"oracle (fp32 torch restatement of the reference CPU path) on {threads} host threads: 1 cold + {n_steady} steady "
                   f"DiT forwards at N=2 (CFG of 1 song), T={T}, L={L} ({t_first:.2f}s / {t_step:.2f}s per step) extrapolated to "
                   f"{args.infer_steps} steps" + (f"; VAE decode of {vae_T} latent frames scaled to {T} ({vae_song_s:.1f}s per song)" if vsd else "")),
        "dit_s_per_step": t_step, "s_per_song": song_s, "logical_cpus": os.cpu_count(),
    }


def dry_run(args, rank, world):
    """The multi-rank control flow of main() with the compute replaced by a memcpy: the same `ace355.dist.run_request` (exact-size
    broadcast, sharding, LM-hint scatter), the same barrier / MAX-over-ranks timing, the same rank-0 JSON line - on CPU tensors over
    gloo.  Exists so that the `--gpus N` launcher and the collective sequence can be exercised where there is no GPU; it measures nothing."""
    import torch.distributed as dist
    from ace355 import dist as a_dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    dev = torch.device("cpu")
    G = args.batch * world if args.scaling == "weak" else args.batch
    s0, s1 = a_dist.shard_range(G, world, rank)
    T, L, D = int(round(args.duration * 25)), args.enc_len, 64
    g = torch.Generator().manual_seed(99)
    enc = torch.randn(1, L, D, generator=g)
    ctx = torch.randn(1, T, 128, generator=g)
    hints = torch.randn(G, T, 64, generator=g)
    seeds = [1000 + i for i in range(G)]
    request = a_dist.pack_request(enc, ctx, seeds, torch.zeros(D), inference_steps=args.infer_steps, guidance_scale=args.guidance) if rank == 0 else None
    ok = True
    seen = {}

    def execute(local):
        nonlocal ok
        b = len(local["seeds"])
        ok = ok and local["range"] == (s0, s1) and local["seeds"] == seeds[s0:s1] and local["global_batch"] == G
        ok = ok and torch.equal(local["encoder_hidden_states"].float(), enc.to(torch.bfloat16).float().expand(b, -1, -1) if world > 1 else enc.expand(b, -1, -1))
        want_ctx = ctx.to(torch.bfloat16).float() if world > 1 else ctx
        got = local["context_latents"].float()
        if args.lm_hints:
            hh = hints[s0:s1].to(torch.bfloat16).float() if world > 1 else hints[s0:s1]
            ok = ok and torch.equal(got[..., :64], hh) and torch.equal(got[..., 64:], want_ctx[..., 64:].expand(b, -1, -1))
        else:
            ok = ok and torch.equal(got, want_ctx.expand(b, -1, -1))
        ok = ok and int(local["knobs"]["inference_steps"]) == args.infer_steps
        seen["b"] = b
        return got.clone()

    def one_pass():
        a_dist.run_request(request, execute, src=0, device=dev, lm_hints=hints if rank == 0 else None, use_lm_hints=args.lm_hints)

    for _ in range(args.warmup):
        one_pass()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    if world > 1:
        dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0, 0.0 if ok else 1.0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher + collectives only)", "dry_run": True, "value": G * args.steps / float(tt[0]),
                          "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling,
                          "collectives_ok": bool(tt[1] == 0), "config": {"global_batch": G, "batch_rank0": seen.get("b", 0), "parallelism": f"dp{world}"}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1) and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; reporting n_gpus = {world}", file=sys.stderr)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the native HIP path has no CPU fallback")
    # ACE355_BENCH_BACKEND=gloo: functional check of the multi-rank flow on a box with fewer GPUs than ranks (ranks share
    # devices round-robin; RCCL refuses two ranks on one GPU).  The driver's runs use the default: one GPU per rank over RCCL.
    backend = os.environ.get("ACE355_BENCH_BACKEND", "nccl")
    device = torch.device(f"cuda:{local_rank % torch.cuda.device_count() if backend != 'nccl' else local_rank}")
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    # Host threads within the container's CPU quota (per rank): torch defaults to half the logical CPUs of the HOST (128 here) while the
    # cgroup grants 16; the OpenMP teams torch.randn / torch.cat spin up for the per-request noise then burn the quota and the kernel
    # throttles the process for up to a 100 ms period with the GPU idle (tools/probe/noise_hiccup_probe.py: randn p99 76.7 ms against a
    # 1.4 ms median; one pass in ~8 took 570-620 ms instead of 475, profiles/r04/r04_bench_invocation_spread_before_gc.txt).
    torch.set_num_threads(max(1, min(torch.get_num_threads(), usable_cpus() // max(1, min(world, torch.cuda.device_count())))))

    import ace355  # noqa: F401
    from ace355 import dist as a_dist
    from ace355.dit import SLOT_COND, SLOT_NULL, prepare_noise, schedule
    from ace355.vae import peak_normalize

    dcfg, vcfg, dit, vae, sd, vsd = build_models(args, device)
    L = args.enc_len
    T = int(round(args.duration * 25))
    S = (T + 1) // 2
    D = dcfg.hidden_size

    # synthetic request: one caption for the batch, per-item seeds; known on rank 0 only
    g = torch.Generator().manual_seed(99)
    enc = torch.randn(L, D, generator=g).to(device)
    null = torch.randn(D, generator=g).to(device)
    ctx_shared = torch.cat([0.5 * torch.randn(T, 64, generator=g), torch.ones(T, 64)], -1).to(device)

    last_local = {}

    def execute(local):
        """One rank's share of a request: per-song noise -> cross-K/V build for the cond / null slots -> sampler -> decode -> peak
        normalise.  The conditioning is resident in HBM; the noise is part of the request as in generate_audio (prepare_noise,
        base.py:1733-1770: one CPU generator per seed - the reference's stream - then one upload), so it is drawn and uploaded in EVERY
        pass, inside the timed region (until round 3 it was cached per seed list outside it)."""
        last_local.clear()
        last_local.update(local)
        tr = [] if trace_passes is not None else None   # ACE355_BENCH_TRACE=1 (diagnostic): host clock + a HIP event after every section

        def mark(name):
            if tr is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                tr.append((name, time.perf_counter(), ev))
        mark("start")
        k = local["knobs"]
        ts = schedule(int(k["inference_steps"]), k["shift"])
        dit.set_condition(SLOT_COND, local["enc_rows"][0])
        dit.set_condition(SLOT_NULL, local["null_condition_emb"].reshape(1, -1), L=L)
        mark("conditions")
        seeds = list(local["seeds"])   # (drawn while the GPU builds the cross-K/V: the order of ace355.dit.generate_latents)
        noise = prepare_noise((len(seeds), T, 64), seeds).to(device, non_blocking=True)
        mark("noise")
        lat = dit.sample(noise, local["context_latents"], ts, guidance_scale=k["guidance_scale"])
        mark("sampler")
        if vae is None:
            if tr is not None:
                trace_passes.append(tr)
            return lat
        wav = vae.decode(lat.transpose(1, 2).contiguous())
        mark("decode")
        out = peak_normalize(wav)
        mark("normalize")
        if tr is not None:
            trace_passes.append(tr)
        return out

    trace_passes = [] if os.environ.get("ACE355_BENCH_TRACE", "0") not in ("", "0") else None

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def measure(scaling, steps, warmup):
        """`steps` timed request passes under one scaling mode -> (global batch, songs of this rank, elapsed MAX over ranks, last output)."""
        G = args.batch * world if scaling == "weak" else args.batch
        s0, s1 = a_dist.shard_range(G, world, rank)
        if G < world:
            raise SystemExit(f"bench.py: global batch {G} over {world} ranks leaves a rank without a song")
        request, hints_all = None, None
        if rank == 0:
            request = a_dist.pack_request(enc[None], ctx_shared[None], [1000 + i for i in range(G)], null, inference_steps=args.infer_steps,
                                          guidance_scale=args.guidance)
            if args.lm_hints:
                hints_all = 0.5 * torch.randn(G, T, 64, generator=g).to(device)
        out = None

        def one_pass():
            return a_dist.run_request(request, execute, src=0, device=device, lm_hints=hints_all, use_lm_hints=args.lm_hints)["local"]

        for _ in range(warmup):
            out = one_pass()
        barrier()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]   # (recorded, never waited on inside the timed region)
        # The cyclic collector stays out of the timed passes: a generation-2 sweep of this process's heap (torch + the reference-shaped
        # state dicts) stops the host for ~55 ms in the middle of a pass with the GPU waiting (ACE355_BENCH_TRACE=1 showed it inside
        # prepare_noise / set_condition at random: profiles/r04/r04_pass_trace.txt) - the interpreter's housekeeping, not work of the request.
        gc.collect()
        gc_was = gc.isenabled()
        if not args.gc_on:
            gc.disable()
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            out = one_pass()
        marks[steps].record()
        barrier()
        elapsed = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        if world > 1:
            tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            elapsed = float(tt.item())
        assert torch.isfinite(out).all(), "non-finite output"
        if trace_passes is not None and rank == 0:
            for n, tr in enumerate(trace_passes[-steps:]):
                host = " ".join(f"{b[0]} {1e3 * (b[1] - a[1]):.1f}" for a, b in zip(tr, tr[1:]))
                gpu = " ".join(f"{b[0]} {a[2].elapsed_time(b[2]):.1f}" for a, b in zip(tr, tr[1:]))
                print(f"[bench trace] pass {n}: host ms: {host} | gpu ms: {gpu} | pass gpu {marks[n].elapsed_time(marks[n + 1]):.1f}", file=sys.stderr)
        per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        step_spread.clear()
        step_spread.update({"min": per[0], "median": per[len(per) // 2], "max": per[-1],
                            "what": "GPU time between the starts of consecutive passes on rank 0's stream (HIP events); the line's ms_per_step is the "
                                    "host clock over all passes, MAX over ranks"})
        return G, s1 - s0, elapsed

    step_spread = {}
    other = None
    if world > 1:   # both modes in one run: the headline one last, so that the profiled pass below re-runs ITS per-rank shape
        o_mode = "weak" if args.scaling == "strong" else "strong"
        o_steps = max(2, args.steps // 4)   # (a side number: it must not double the run the driver times around the headline K steps)
        oG, oB, o_el = measure(o_mode, o_steps, max(1, args.warmup // 2))
        other = {"scaling": o_mode, "value": oG * o_steps / o_el, "unit": "songs/s", "ms_per_step": 1000.0 * o_el / o_steps,
                 "global_batch": oG, "batch_rank0": oB, "steps": o_steps}
    G, B, elapsed = measure(args.scaling, args.steps, args.warmup)

    songs = G * args.steps
    value = songs / elapsed
    result = {
        "metric": (f"songs/sec ({args.duration:g} s audio @ {args.infer_steps} DiT steps, CFG {args.guidance:g} + APG, batch {args.batch} "
                   + ("per GPU" if args.scaling == "weak" and world > 1 else "in total") + (", DiT + VAE decode)" if not args.no_vae else ", DiT-only)")),
        "value": value, "unit": "songs/s", "rtf": value * args.duration, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps, "step_ms_spread": dict(step_spread), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "mxfp8 (four big projections) + bf16" if args.fp8 else "bf16", "data": "synthetic",
        "config": {"workload": f"acestep-5Hz base DiT (24L/2048d, 1.575B params, random init) + Oobleck decoder, {args.duration:g} s audio "
                               f"(T={T}), {args.infer_steps} steps, CFG {args.guidance:g} (2x{B} sequences/forward on rank 0), L={L}, "
                               f"{G} songs per request over {world} GPU(s)" + (", DiT-only" if args.no_vae else "") + (", per-item LM hints scattered" if args.lm_hints else ""),
                   "audio_seconds": args.duration, "infer_steps": args.infer_steps, "batch_per_gpu": B, "batch_rank0": B, "global_batch": G,
                   "parallelism": f"dp{world}", "tiny": bool(args.tiny), "gc_disabled_in_timed_region": not args.gc_on,
                   "sampler_chains_per_gpu": (dit.dual_count() > 0) + 1 if hasattr(dit, "dual_count") else 1,
                   # the bench keeps the library's default (fastest) launch policy on every rank: a song's low bits then follow the size of the
                   # slice it runs in (~3e-3 rel L2, both at the reference's distance).  NativeHandler.generate_music(data_parallel=True) selects
                   # the launch-shape-independent mode instead (same bits on 1 / 2 / 4 / 8 GPUs; tests/test_dist_gpu.py)
                   "batch_dependent_bits": True},
    }
    if other is not None:
        result[other["scaling"]] = other

    def one_pass(collective=True):
        # the rank-0-only profiled passes after the timed region must not enter a collective the other ranks never join: they
        # re-run this rank's share of the last timed request
        assert not collective
        return execute(dict(last_local))

    if rank == 0 and not args.no_roofline:
        # dominant kernel = gemm_kernel: algorithmic FLOPs per launch / HIP-event launch time, one extra profiled pass
        dit.set_profile(True)
        if vae is not None:
            vae.set_profile(True)
        dual0 = dit.dual_count() if hasattr(dit, "dual_count") else 0
        one_pass(collective=False)
        torch.cuda.synchronize()
        p = dit.get_profile()
        dit.set_profile(False)
        # Two chains (the default for >= 2 songs): the DiT launches of the pass ran on two hardware queues, each planned for 128 of the
        # 256 CUs.  ace355_dit_get_profile reports CHIP time (the sum of the launch durations over both chains / 2); the average
        # launch duration a kernel trace shows is the per-launch figure: chip time x chains / launches.
        chains = 2 if hasattr(dit, "dual_count") and dit.dual_count() > dual0 else 1
        gemm_tf = p["gemm_flops"] / (p["gemm_ms"] * 1e-3) / 1e12 if p["gemm_ms"] > 0 else 0.0
        peak = PEAK_MXFP8_TFLOPS if args.fp8 else PEAK_BF16_TFLOPS
        kname = ("gemm_sp_kernel<.., FP8> (MX-scaled fp8 MFMA 32x32x64 for QKV / o_proj / gate|up / down, 192x256x128 tiles; the cross-attention "
                 "projections and small launches stay on the bf16 kernel: `achieved` averages over all GEMM launches)") if args.fp8 else \
            "gemm_sp_kernel (bf16 MFMA 16x16x32; 192x256x64 and 192x128x64 8-wave / 128-192x128x64 4-wave tiles, LDS-DMA staging)"
        result["roofline"] = {"bound": "mfma", "kernel": kname,
                              "achieved": gemm_tf, "peak": peak, "unit": "TFLOP/s", "frac": gemm_tf / peak,
                              "traffic": None, "gemm_ms_per_pass": p["gemm_ms"], "gemm_launches_per_pass": p["gemm_launches"],
                              "avg_launch_us": 1000.0 * p["gemm_ms"] * chains / max(p["gemm_launches"], 1),
                              "flops_per_launch": p["gemm_flops"] / max(p["gemm_launches"], 1),
                              "concurrent_chains": chains, "cus_per_launch": 256 // chains,
                              "accounting": ("achieved = flops_per_launch / avg_launch_us x concurrent_chains: with two chains every launch is planned "
                                             "for 128 CUs and two launches run side by side, so a launch's rate is priced against half the chip's peak "
                                             "(equivalently: all GEMM flops of the pass / (sum of launch durations / 2))") if chains == 2 else
                                            "achieved = flops_per_launch / avg_launch_us (one chain: every launch has the whole chip)",
                              "attn_tflops": p["attn_flops"] / (p["attn_ms"] * 1e-3) / 1e12 if p["attn_ms"] > 0 else 0.0,
                              "attn_ms_per_pass": p["attn_ms"]}
        if vae is not None:
            vp = vae.get_profile()
            vae.set_profile(False)
            result["roofline"]["vae_conv_tflops"] = vp["conv_flops"] / (vp["conv_ms"] * 1e-3) / 1e12 if vp["conv_ms"] > 0 else 0.0
            result["roofline"]["vae_conv_ms_per_pass"] = vp["conv_ms"]
        if not args.fp8 and hasattr(dit, "set_norm_fold"):
            # The GEMM launches of the default path also do the RMSNorm work of 71 of the 72 norms per forward (folded into their
            # epilogues); the same launches without that work, for comparison with earlier rounds' fractions:
            dit.set_norm_fold(False)
            dit.set_profile(True)
            one_pass(collective=False)
            torch.cuda.synchronize()
            q = dit.get_profile()
            dit.set_profile(False)
            dit.set_norm_fold(True)
            q_tf = q["gemm_flops"] / (q["gemm_ms"] * 1e-3) / 1e12 if q["gemm_ms"] > 0 else 0.0
            result["roofline"]["norm_fold"] = {
                "default": "on: the residual GEMMs also write bf16(h*g) + row sums, the QKV / cross-q / gate|up GEMMs apply rstd + shift W^T; "
                           "the standalone rmsnorm kernels (13.0 + 13.0 + 7.3 us per layer) are gone from the pass",
                "gemm_ms_per_pass_norms_as_kernels": q["gemm_ms"], "achieved_norms_as_kernels": q_tf, "frac_norms_as_kernels": q_tf / peak}
        # L2-miss-side bytes / GB/s of the dominant kernel = PMC bytes per launch (committed profile of this command) / live launch time;
        # only when that profile was taken on the library being timed (same source sha), else it is named as stale and traffic stays null
        pm = pmc_traffic_bytes_per_launch()
        result["roofline"]["library_src_sha"] = library_source_sha()
        if pm is not None:
            src = (f"{pm['file']} + WRITE_SIZE twin (commit {pm['commit']}, lib_src_sha {pm['lib_src_sha']}): rocprofv3 --pmc passes of this command, "
                   "committed (not collected live); fabric-side of the L2s, so Infinity-Cache hits are included: an upper bound on HBM bytes")
            if pm["current"]:
                result["roofline"]["traffic"] = pm["bytes"]
                result["roofline"]["hbm_gbps"] = pm["bytes"] * chains / (result["roofline"]["avg_launch_us"] * 1e-6) / 1e9
                result["roofline"]["traffic_source"] = src
            else:
                result["roofline"]["hbm_gbps"] = None
                result["roofline"]["traffic_stale"] = {"bytes": pm["bytes"], "source": src,
                                                       "why": "the newest PMC profile under profiles/ was not taken on the library sources in this tree"}
        else:
            result["roofline"]["hbm_gbps"] = None
        do_cfg = args.guidance > 1.0
        alg = B * ((2 if do_cfg else 1) * args.infer_steps * dit_flops_per_forward_per_seq(dcfg, S, L) + (0 if args.no_vae else T * vae_flops_per_frame(vcfg)))
        result["algorithmic_tflop_per_step"] = alg / 1e12
        # SURVEY 8d counts the un-shortcut algorithm (CFG null branch cross-attention included); the path EXECUTES less: the
        # null branch's cross-attention is the exact constant it is.  Both rates are reported (this rank's songs / wall time).
        result["achieved_tflops_whole_path"] = alg / 1e12 / (elapsed / args.steps)
        executed = p["gemm_flops"] + p["attn_flops"] + (vp["conv_flops"] if vae is not None else 0.0)
        result["executed_tflop_per_step"] = executed / 1e12
        result["achieved_tflops_executed"] = executed / 1e12 / (elapsed / args.steps)
        result["null_branch_shortcut_tflop_saved"] = (alg - executed) / 1e12

    if rank == 0 and not args.no_roofline:
        # board state while the DiT part of one more pass is in flight (rocm-smi, best effort): the same binary measures 493-549 ms per pass
        # across the boxes of the pool, and sclk / package power under load say which kind of box a line came from
        try:
            import re
            import subprocess
            dit.set_condition(SLOT_COND, enc)
            dit.set_condition(SLOT_NULL, null.reshape(1, -1), L=L)
            _ = dit.sample(prepare_noise((B, T, 64), list(last_local["seeds"])).to(device), ctx_shared[None].expand(B, -1, -1).contiguous(), schedule(args.infer_steps, 1.0),
                           guidance_scale=args.guidance)   # queued, not awaited
            out_smi = subprocess.run(["rocm-smi", "-d", str(device.index or 0), "--showclocks", "--showpower"], capture_output=True, text=True,
                                     timeout=20).stdout
            torch.cuda.synchronize()
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out_smi)
            pw = re.search(r"Power \(W\): ([0-9.]+)", out_smi)
            result["gpu_state_under_load"] = {"sclk_mhz": int(sclk.group(1)) if sclk else None, "package_power_w": float(pw.group(1)) if pw else None,
                                              "source": "rocm-smi during one extra untimed DiT pass"}
        except Exception as e:  # no rocm-smi, no permission, ...: the line is complete without it
            result["gpu_state_under_load"] = {"error": str(e)[:120]}

    if rank == 0 and not args.no_roofline:
        # what THIS board's matrix pipe does on a pure MFMA loop with random bf16 operands (no memory traffic), right after the passes:
        # the same binary measures 493-549 ms per pass over the boxes of the pool (DVFS under one power cap), and a line is only
        # comparable with another box's beside this number (nominal dense peak 2500; round 2's reference box: 1964)
        try:
            import ctypes as C
            from ace355 import native
            tf = C.c_double()
            torch.cuda.synchronize()
            native.check(native.lib().ace355_box_probe_mfma(150000, C.byref(tf)), "box_probe_mfma")
            result["box_probe"] = {"mfma_random_bf16_tflops": tf.value, "what": "pure v_mfma_f32_32x32x16_bf16 loop, random operands, 8 waves per CU, "
                                   "no memory traffic (ace355_box_probe_mfma)", "gemm_achieved_over_probe": result["roofline"]["achieved"] / tf.value if tf.value else None}
        except Exception as e:
            result["box_probe"] = {"error": str(e)[:160]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # reported at N=1 only
        result["cpu_baseline"] = cpu_baseline(args, dcfg, vcfg, sd, vsd, enc.cpu(), null.cpu(), ctx_shared.cpu()[None], T, L)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
