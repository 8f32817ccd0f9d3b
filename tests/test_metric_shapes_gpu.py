"""GPU parity AT THE SHAPES THE HEADLINE RUNS (BASELINE.json metric: 30 s, batch 8, CFG => N = 16 sequences, M = 6000 token rows,
Mc = 3000 cross-attention rows) and at configs[2] (120 s, S = 1500), against vectors captured from the imported reference
(tests/golden/make_golden.py g11 / g12 / g13, full-size architecture, fp32 CPU).  These are the launches bench.py times:
192x256 persistent tiles, the 192x128 mid tile, the residual / SwiGLU / head-norm epilogues at M = 6000, attn3_kernel<4> at
N = 16 and attn3_kernel<8> at S = 1500.  Inputs are regenerated from seeds (CPU generators) and pinned by checksums.

Gates: twice the distance the oracle itself shows from the fixture's fp32 result when it stores weights and
contraction operands in bf16 - measured per fixture by tests/golden/make_drift.py (the oracle needs minutes to hours for these shapes) and read
from tests/golden/bf16_storage_drift.json (tests/_drift.py).  Until round 6 they were "2-3 x what was measured".
"""
import os

import numpy as np
import pytest
import torch

import _drift

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _inputs(B, T, seed0=1000, ctx_seed=45):
    """Same construction as make_golden.metric_inputs (prepare_noise per-item CPU generators, base.py:1733-1770)."""
    from ace355.dit import prepare_noise
    x = prepare_noise((B, T, 64), [seed0 + i for i in range(B)])
    g = torch.Generator().manual_seed(ctx_seed)
    ctx1 = torch.cat([0.5 * torch.randn(1, T, 64, generator=g), torch.ones(1, T, 64)], -1)
    return x, ctx1


def _close(a, b):
    return abs(a - b) <= 1e-9 * abs(b)



def test_metric_shape_forward_vs_reference_golden(gpu_device, golden_dir, full_dit_seed4):
    """G11: AceStepDiTModel.forward (base.py:1303-1507) at N = 16 (8 conditional + 8 null_condition_emb.expand_as, :1905-1911),
    T = 750, L = 769: velocity of all 16 sequences + residual-stream taps after layers 0 and 23."""
    G = np.load(f"{golden_dir}/g11_metric_forward.npz")
    dit, cfg, null, wsum = full_dit_seed4
    assert abs(wsum - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    B, T = 8, 750
    x8, ctx1 = _inputs(B, T)
    assert _close(float(x8.double().abs().sum()), float(G["x_sum"])) and _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
    x = torch.cat([x8, x8], 0)
    ctx = ctx1.expand(2 * B, -1, -1).contiguous()
    dit.set_condition(0, enc[0])
    dit.set_condition(1, null.reshape(1, -1), L=enc.shape[1])
    S, N = T // 2, 2 * B
    taps = {li: torch.empty(N * S, cfg.hidden_size, device=gpu_device) for li in (0, 23)}
    for li, buf in taps.items():
        dit.set_tap(li, buf)
    try:
        t = [float(G["t"])] * N
        v = dit.forward(x, ctx, t, t, [0] * B + [1] * B)
        torch.cuda.synchronize()
        tap23 = taps[23].clone()
        v_again = dit.forward(x, ctx, t, t, [0] * B + [1] * B)
        torch.cuda.synchronize()
        assert torch.equal(v, v_again) and torch.equal(tap23, taps[23]), "the same forward twice: not bit-identical"
    finally:
        for li in taps:
            dit.set_tap(li, None)
    ref = torch.from_numpy(G["v"])
    seqs, stride = G["tap_seqs"].tolist(), int(G["tap_stride"])
    r = _rel(v, ref)
    r_c, r_u = _rel(v[:B], ref[:B]), _rel(v[B:], ref[B:])
    r0 = _rel(taps[0].view(N, S, -1)[seqs][:, ::stride], torch.from_numpy(G["l0_out"]))
    r23 = _rel(taps[23].view(N, S, -1)[seqs][:, ::stride], torch.from_numpy(G["l23_out"]))
    print(f"metric-shape forward (N=16, T=750): rel L2 vs reference fp32 = {r:.3e} (cond {r_c:.3e}, null {r_u:.3e}); "
          f"taps layer 0 {r0:.3e}, layer 23 {r23:.3e}")
    assert torch.isfinite(v).all()
    # measured: v 5.8e-3 (cond 6.2e-3, null 5.1e-3), taps 3.8e-3 / 4.1e-3
    for what, val, key in (("v", r, "v"), ("v, conditional half", r_c, "v_cond"), ("v, null half", r_u, "v_null"), ("layer-0 tap", r0, "l0"), ("layer-23 tap", r23, "l23")):
        _drift.check(f"G11 {what}", val, _drift.table("g11", key))


def test_metric_batch_sampler_vs_reference_golden(gpu_device, golden_dir, full_dit_seed4):
    """G12: generate_audio (base.py:1783-1989) at the metric batch - 8 songs x 30 s, CFG 7 + APG, 3 steps - through
    ace355_dit_sample: the exact launch sequence bench.py times (shared timestep row, null-branch shortcut, 192x256 tiles)."""
    from ace355.dit import generate_latents
    G = np.load(f"{golden_dir}/g12_metric_sampler.npz")
    dit, cfg, null, wsum = full_dit_seed4
    assert abs(wsum - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    B, T = 8, 750
    _, ctx1 = _inputs(B, T)
    assert _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
    out = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), seed=G["seeds"].tolist(),
                           infer_steps=int(G["steps"]), diffusion_guidance_sale=float(G["guidance"]))["target_latents"]
    ref = torch.from_numpy(G["out"])
    r = _rel(out, ref)
    per = [_rel(out[i], ref[i]) for i in range(B)]
    print(f"metric-batch sampler (B=8, 3 steps, CFG 7 + APG): rel L2 vs reference fp32 = {r:.3e}; per item max {max(per):.3e}")
    _drift.check("G12 all songs", r, _drift.table("g12", "out"))   # measured 4.0e-3
    _drift.check("G12 worst song", max(per), _drift.table("g12", "per_song_max"))


def test_a_song_does_not_depend_on_its_batch_in_the_shape_independent_mode(gpu_device, golden_dir, full_dit_seed4):
    """VERDICT r5 weak 3 / ADVICE r4: "1 GPU vs 8 GPUs gives the same song".  G12's request (8 songs x 30 s, CFG 7 + APG, 3 steps) as ONE call,
    as calls of 4 + 4, 2 + ... and of single songs - the 1 / 2 / 4 / 8-GPU shares of the metric batch - in the library's
    launch-shape-independent mode (`ace355_gemm_set_k_rotation(0)`: no K rotation, no split-K, no split-KV / key-split attention; one sampler
    chain; the folded RMSNorms stay on), which the data-parallel handler path selects by default: every song must come out bit for bit the same whatever call it was part of,
    and still match the reference.  With the default (fastest) policy the same comparison gives ~3e-3 - printed for the record."""
    from ace355 import native
    from ace355.dit import generate_latents
    G = np.load(f"{golden_dir}/g12_metric_sampler.npz")
    dit, cfg, null, wsum = full_dit_seed4
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    B, T = 8, 750
    _, ctx1 = _inputs(B, T)
    seeds = G["seeds"].tolist()
    ref = torch.from_numpy(G["out"])

    def run(items):
        n = len(items)
        return generate_latents(dit, null, enc.expand(n, -1, -1), ctx1.expand(n, -1, -1).contiguous(), seed=[seeds[i] for i in items],
                                infer_steps=int(G["steps"]), diffusion_guidance_sale=float(G["guidance"]))["target_latents"].cpu()

    fast8, fast1 = run(range(8)), run([5])
    krot = native.gemm_set_k_rotation(0)
    dual = dit.set_dual(0)
    try:
        all8 = run(range(8))
        halves = torch.cat([run(range(0, 4)), run(range(4, 8))])
        pairs = torch.cat([run([0, 1]), run([2, 3]), run([4, 5]), run([6, 7])])
        singles = torch.cat([run([i]) for i in (0, 5, 7)])
    finally:
        native.gemm_set_k_rotation(krot)
        dit.set_dual(dual)
    r = _rel(all8, ref)
    print(f"shape-independent mode (B=8, 3 steps): vs reference fp32 {r:.3e}; 8 == 4+4: {torch.equal(all8, halves)}, == 2+2+2+2: {torch.equal(all8, pairs)}, "
          f"songs 0 / 5 / 7 alone == inside the batch: {torch.equal(singles, all8[[0, 5, 7]])}; default policy: song 5 alone vs in the batch {_rel(fast1, fast8[5:6]):.3e}")
    _drift.check("G12 in the shape-independent mode", r, _drift.table("g12", "out"))
    assert torch.equal(all8, halves) and torch.equal(all8, pairs) and torch.equal(singles, all8[[0, 5, 7]])


def test_metric_batch_sampler_norm_fold_on_off(gpu_device, golden_dir, full_dit_seed4):
    """The folded RMSNorm path (default in big-M sampler calls: GemmEpilogue nf_* / nc_*, include/ace355.h ace355_dit_set_norm_fold)
    against the same call with the norms as kernels of their own, and both against the reference golden G12."""
    from ace355.dit import generate_latents
    G = np.load(f"{golden_dir}/g12_metric_sampler.npz")
    dit, cfg, null, wsum = full_dit_seed4
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    B, T = 8, 750
    _, ctx1 = _inputs(B, T)
    ref = torch.from_numpy(G["out"])
    outs = {}
    try:
        for fold in (True, False):
            dit.set_norm_fold(fold)
            outs[fold] = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), seed=G["seeds"].tolist(),
                                          infer_steps=int(G["steps"]), diffusion_guidance_sale=float(G["guidance"]))["target_latents"].cpu()
    finally:
        dit.set_norm_fold(True)
    r_on, r_off, r_ab = _rel(outs[True], ref), _rel(outs[False], ref), _rel(outs[True], outs[False])
    print(f"norm fold: on vs reference {r_on:.3e}, off vs reference {r_off:.3e}, on vs off {r_ab:.3e}")
    assert torch.isfinite(outs[True]).all()
    _drift.check("G12 folded norms", r_on, _drift.table("g12", "out"))
    _drift.check("G12 norms as kernels", r_off, _drift.table("g12", "out"))
    assert r_ab < 6e-3, r_ab   # (native vs native)
    assert not torch.equal(outs[True], outs[False])  # the two paths really are different launch sequences


def test_dual_chain_sampler_contract(gpu_device, golden_dir, full_dit_seed4):
    """Dual-chain sampler (include/ace355.h ace355_dit_set_dual; the songs of generate_audio are independent, base.py:1783-1989): the
    metric batch (G12: 8 songs x 30 s, CFG 7 + APG, 3 steps) FORCED to run as two half-batch samplers on two hardware queues (mode 2;
    the default policy splits requests of 2-4 songs only).  Contract: (1) the call really ran as two chains; (2) each song's result
    is BIT-identical to what a one-chain call with that song's half-batch alone returns (songs 0-3 / 4-7), eagerly and as a
    replayed graph; (3) the result matches the reference golden like the one-chain path does; (4) an odd batch (5 songs: 3 + 2) and
    a batch of 2 (1 + 1; split by the DEFAULT policy) obey (2)."""
    from ace355.dit import generate_latents
    G = np.load(f"{golden_dir}/g12_metric_sampler.npz")
    dit, cfg, null, wsum = full_dit_seed4
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    B, T = 8, 750
    _, ctx1 = _inputs(B, T)
    ref = torch.from_numpy(G["out"])
    steps, seeds = int(G["steps"]), G["seeds"].tolist()

    def run(items):
        b = len(items)
        return generate_latents(dit, null, enc.expand(b, -1, -1), ctx1.expand(b, -1, -1).contiguous(), seed=[seeds[i] for i in items],
                                infer_steps=steps, diffusion_guidance_sale=float(G["guidance"]))["target_latents"].cpu()
    try:
        dit.set_dual(1)
        n0 = dit.dual_count()
        d2 = run(range(2))
        if dit.dual_count() != n0 + 1:
            # The product falls back to one chain (with a note on stderr) when no side stream can be put on a hardware queue of its own
            # - a property of the runtime's queue pool in this process, not of the path.  Then the fallback itself is what is checked.
            dit.set_dual(0)
            assert torch.equal(d2, run(range(2))), "one-chain fallback of the dual-chain sampler differs from the one-chain path"
            pytest.skip("no side stream on a hardware queue of its own in this process: the dual-chain sampler ran (correctly) as one chain")
        run(range(8))
        assert dit.dual_count() == n0 + 1, "the default policy must keep the metric batch on one chain"
        dit.set_dual(2)
        dual = run(range(8))
        assert dit.dual_count() == n0 + 2
        again = run(range(8))
        assert torch.equal(dual, again)
        dit.set_graph(True)
        try:
            g1, g2 = run(range(8)), run(range(8))
            assert dit.graph_stats()["replays"] >= 1
        finally:
            dit.set_graph(False)
        assert torch.equal(dual, g1) and torch.equal(dual, g2), "graph replay of the two-chain sequence differs"
        d5 = run(range(5))
        dit.set_dual(0)
        n1 = dit.dual_count()
        single = run(range(8))
        halves = torch.cat([run(range(0, 4)), run(range(4, 8))], 0)
        assert dit.dual_count() == n1
        s5 = torch.cat([run(range(0, 3)), run(range(3, 5))], 0)
        s2 = torch.cat([run([0]), run([1])], 0)
    finally:
        dit.set_dual(1)
    r_d, r_s, r_ds = _rel(dual, ref), _rel(single, ref), _rel(dual, single)
    print(f"dual-chain sampler (B=8, 3 steps): two chains vs reference fp32 {r_d:.3e}, one chain vs reference {r_s:.3e}, two chains vs one {r_ds:.3e}; "
          f"two chains == one-chain calls on the half-batches: {torch.equal(dual, halves)} (B=8), {torch.equal(d5, s5)} (B=5), {torch.equal(d2, s2)} (B=2)")
    assert torch.equal(dual, halves) and torch.equal(d5, s5) and torch.equal(d2, s2)
    _drift.check("G12 two sampler chains", r_d, _drift.table("g12", "out"))
    _drift.check("G12 one sampler chain", r_s, _drift.table("g12", "out"))
    assert r_ds < 6e-3, r_ds   # (native vs native)


def test_full_schedule_sampler_vs_reference_golden(gpu_device, golden_dir, full_dit_seed4):
    """G15: the reference's generate_audio over the WHOLE 27-step schedule at full size (B = 3 x 30 s, CFG 7 + APG: N = 6 sequences,
    2250 token rows = big GEMM tiles + folded RMSNorm in the native sampler) against one ace355_dit_sample call, folded and unfolded."""
    from ace355.dit import generate_latents
    G = np.load(f"{golden_dir}/g15_full_schedule_sampler.npz")
    dit, cfg, null, wsum = full_dit_seed4
    assert abs(wsum - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    B, T = 3, 750
    _, ctx1 = _inputs(B, T)
    assert _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
    ref = torch.from_numpy(G["out"])
    res = {}
    try:
        for fold in (True, False):
            dit.set_norm_fold(fold)
            out = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), seed=G["seeds"].tolist(),
                                   infer_steps=int(G["steps"]), diffusion_guidance_sale=float(G["guidance"]))["target_latents"].cpu()
            res[fold] = (_rel(out, ref), max(_rel(out[i], ref[i]) for i in range(B)))
            assert torch.isfinite(out).all()
    finally:
        dit.set_norm_fold(True)
    print(f"full-schedule sampler (B=3, 27 steps, CFG 7 + APG) vs reference fp32: folded {res[True][0]:.3e} (per item max {res[True][1]:.3e}), "
          f"norms as kernels {res[False][0]:.3e} (per item max {res[False][1]:.3e})")
    for fold in (True, False):   # measured 3.47e-3 (folded and unfolded alike)
        _drift.check(f"G15 fold {fold}", res[fold][0], _drift.table("g15", "out"))
        _drift.check(f"G15 fold {fold}, worst song", res[fold][1], _drift.table("g15", "per_song_max"))


def test_bench_request_sampler_vs_reference_golden(gpu_device, golden_dir, full_dit_seed4):
    """G16: the EXACT request bench.py times - 8 songs x 30 s, CFG 7 + APG, the whole 27-step schedule at full size (N = 16 sequences,
    6000 token rows: 192x256 persistent tiles, folded RMSNorm, 12-wave attention) - against the reference's generate_audio
    (677 s of reference CPU time), folded and with the norms as kernels."""
    from ace355.dit import generate_latents
    G = np.load(f"{golden_dir}/g16_bench_request_sampler.npz")
    dit, cfg, null, wsum = full_dit_seed4
    assert abs(wsum - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    B, T = 8, 750
    _, ctx1 = _inputs(B, T)
    assert _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
    ref = torch.from_numpy(G["out"])
    res = {}
    try:
        for fold in (True, False):
            dit.set_norm_fold(fold)
            out = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), seed=G["seeds"].tolist(),
                                   infer_steps=int(G["steps"]), diffusion_guidance_sale=float(G["guidance"]))["target_latents"].cpu()
            res[fold] = (_rel(out, ref), max(_rel(out[i], ref[i]) for i in range(B)))
            assert torch.isfinite(out).all()
    finally:
        dit.set_norm_fold(True)
    print(f"bench-request sampler (B=8, 27 steps, CFG 7 + APG) vs reference fp32: folded {res[True][0]:.3e} (per item max {res[True][1]:.3e}), "
          f"norms as kernels {res[False][0]:.3e} (per item max {res[False][1]:.3e})")
    for fold in (True, False):   # measured 3.48e-3
        _drift.check(f"G16 (the bench's request) fold {fold}", res[fold][0], _drift.table("g16", "out"))
        _drift.check(f"G16 fold {fold}, worst song", res[fold][1], _drift.table("g16", "per_song_max"))


@pytest.mark.parametrize("B", [1, 2])
def test_small_batch_request_vs_reference_golden(gpu_device, golden_dir, full_dit_seed4, B):
    """BASELINE configs[1] (30 s, 27 steps, batch 1) and the 2-song share of a 4-GPU split of the metric batch: the small-M
    launch regime (M = 750 / 1500 token rows: 4-wave deep-pipeline tiles, ordered split-K, split-KV attention).  Songs are independent
    (per-item seeds, one caption), so items 0.. of G16 ARE the reference's answer for the smaller request; run twice: bit-reproducible."""
    from ace355.dit import generate_latents
    G = np.load(f"{golden_dir}/g16_bench_request_sampler.npz")
    dit, cfg, null, wsum = full_dit_seed4
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    T = 750
    _, ctx1 = _inputs(B, T)
    ref = torch.from_numpy(G["out"])[:B]
    kw = dict(seed=G["seeds"].tolist()[:B], infer_steps=int(G["steps"]), diffusion_guidance_sale=float(G["guidance"]))
    a = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), **kw)["target_latents"].cpu()
    b = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), **kw)["target_latents"].cpu()
    r = _rel(a, ref)
    print(f"batch-{B} request (27 steps, CFG 7 + APG) vs the reference's items of G16: rel L2 {r:.3e}")
    assert torch.isfinite(a).all()
    _drift.check(f"first {B} song(s) of G16 as their own request", r, _drift.slice_drift("g16", B))
    assert torch.equal(a, b), "same request twice must be bit-identical"


def test_120s_forward_vs_reference_golden_and_batch16(gpu_device, golden_dir, full_dit_seed4):
    """G13 / BASELINE configs[2] (120 s, T = 3000, S = 1500): a CFG pair vs the reference (N = 2: the 96-row attention launches, i.e. the
    key-split attn_rot_kernel by default), then the same pair inside a batch of N = 16, where launch_attention takes the 192-row
    attn_gqa_kernel<6> and the GEMMs M = 24000 rows: both runs must match the reference; with ace355_gemm_set_k_rotation(0) (one summation
    order whatever the launch shape, GEMM K order and attention kernel family alike) they must agree with each other exactly."""
    G = np.load(f"{golden_dir}/g13_120s_forward.npz")
    dit, cfg, null, wsum = full_dit_seed4
    assert abs(wsum - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    T = 3000
    x1, ctx1 = _inputs(1, T, seed0=2000, ctx_seed=46)
    assert _close(float(x1.double().abs().sum()), float(G["x_sum"])) and _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
    dit.set_condition(0, enc[0])
    dit.set_condition(1, null.reshape(1, -1), L=enc.shape[1])
    t = float(G["t"])
    ref = torch.from_numpy(G["v"])
    S = T // 2
    tap = torch.empty(2 * S, cfg.hidden_size, device=gpu_device)
    dit.set_tap(23, tap)
    try:
        v2 = dit.forward(torch.cat([x1, x1]), ctx1.expand(2, -1, -1).contiguous(), [t, t], [t, t], [0, 1])
        torch.cuda.synchronize()
    finally:
        dit.set_tap(23, None)
    assert torch.equal(v2, dit.forward(torch.cat([x1, x1]), ctx1.expand(2, -1, -1).contiguous(), [t, t], [t, t], [0, 1])), "not bit-reproducible"
    r2 = _rel(v2, ref)
    r23 = _rel(tap.view(2, S, -1)[:, ::100], torch.from_numpy(G["l23_out"]))
    # batch of 16: item 3 (cond) and item 11 (null) carry the golden pair, the rest other seeds
    xo, _ = _inputs(8, T, seed0=3000, ctx_seed=46)
    xo[3] = x1[0]
    v16 = dit.forward(torch.cat([xo, xo]), ctx1.expand(16, -1, -1).contiguous(), [t] * 16, [t] * 16, [0] * 8 + [1] * 8)
    pair = torch.stack([v16[3], v16[11]])
    r16, rx = _rel(pair, ref), _rel(pair, v2)
    # ... and with ONE K order whatever the launch shape (include/ace355.h ace355_gemm_set_k_rotation(0)): the same arithmetic per wave
    # whatever the block / tile shape, measured 0.0 between the two runs
    from ace355 import native
    krot = native.gemm_set_k_rotation(0)
    try:
        v2n = dit.forward(torch.cat([x1, x1]), ctx1.expand(2, -1, -1).contiguous(), [t, t], [t, t], [0, 1])
        v16n = dit.forward(torch.cat([xo, xo]), ctx1.expand(16, -1, -1).contiguous(), [t] * 16, [t] * 16, [0] * 8 + [1] * 8)
    finally:
        native.gemm_set_k_rotation(krot)
    rxn = _rel(torch.stack([v16n[3], v16n[11]]), v2n)
    print(f"120 s forward: N=2 vs reference {r2:.3e} (layer-23 tap {r23:.3e}); inside N=16 (attn3_kernel<8>) vs reference {r16:.3e}, "
          f"vs the N=2 run {rx:.3e} (K rotation mode {krot}), {rxn:.3e} with the rotation off ({_rel(v2n, ref):.3e} vs the reference)")
    assert torch.isfinite(v16).all()
    _drift.check("G13 pair alone", r2, _drift.table("g13", "v"))   # measured 5.9e-3, 5.9e-3, 4.2e-3
    _drift.check("G13 pair inside N = 16", r16, _drift.table("g13", "v"))
    _drift.check("G13 layer-23 tap", r23, _drift.table("g13", "l23"))
    # The default K rotation and the key-split attention kernel of the 96-row launches sum
    # in an order that depends on the launch shape: 3.2e-3 between the two runs, both at the reference's distance; with mode 0 the two
    # runs agree exactly (measured 0.0).
    assert rx < (5e-3 if krot else 2e-3), rx
    assert rxn == 0.0, rxn
    _drift.check("G13 pair, K rotation off", _rel(v2n, ref), _drift.table("g13", "v"))
    assert _rel(v16[2], v16[3]) > 0.3  # other seeds really are other songs


def test_240s_forward_vs_reference_golden(gpu_device, golden_dir, full_dit_seed4):
    """G14 / BASELINE configs[3] shape (240 s, T = 6000, S = 3000; each of the 8 GPUs runs 4 such songs x 60 steps): one CFG pair through
    the native forward vs the imported reference."""
    G = np.load(f"{golden_dir}/g14_240s_forward.npz")
    dit, cfg, null, wsum = full_dit_seed4
    assert abs(wsum - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    T = 6000
    x1, ctx1 = _inputs(1, T, seed0=4000, ctx_seed=47)
    assert _close(float(x1.double().abs().sum()), float(G["x_sum"])) and _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
    dit.set_condition(0, enc[0])
    dit.set_condition(1, null.reshape(1, -1), L=enc.shape[1])
    t = float(G["t"])
    v = dit.forward(torch.cat([x1, x1]), ctx1.expand(2, -1, -1).contiguous(), [t, t], [t, t], [0, 1])
    assert torch.equal(v, dit.forward(torch.cat([x1, x1]), ctx1.expand(2, -1, -1).contiguous(), [t, t], [t, t], [0, 1])), "not bit-reproducible"
    r = _rel(v, torch.from_numpy(G["v"]))
    print(f"240 s forward (N=2, T=6000): rel L2 vs reference fp32 = {r:.3e}")
    assert torch.isfinite(v).all()
    _drift.check("G14 (240 s)", r, _drift.table("g14", "v"))


@pytest.mark.parametrize("name,T,B,steps,precision", [
    ("configs[2]: 120 s, batch 8 (N = 16 sequences, 24000 token rows)", 3000, 8, 2, "bf16"),
    ("configs[3] per-GPU share: 240 s, batch 4", 6000, 4, 2, "bf16"),
    ("configs[4] per-GPU shape: 600 s, batch 8, fp8 MFMA (N = 16 sequences, 120000 token rows)", 15000, 8, 2, "mxfp8")])
def test_long_config_sampler_properties(gpu_device, golden_dir, full_dit_seed4, name, T, B, steps, precision):
    """The two 8-GPU configurations of BASELINE.json at their per-GPU shapes (the fp32 oracle would need minutes per step there):
    size-independent properties of the sampler - finite, deterministic (same request twice is bit-identical), songs independent
    (item 1 of the batch equals the same seed run alone up to fp32 summation order across tile shapes), seeds matter."""
    from ace355.dit import generate_latents
    dit, cfg, null, wsum = full_dit_seed4
    enc = torch.from_numpy(np.load(f"{golden_dir}/g4_full_forward.npz")["enc"])
    _, ctx1 = _inputs(1, T, seed0=5000, ctx_seed=48)
    kw = dict(infer_steps=steps, diffusion_guidance_sale=7.0)
    seeds = [7000 + i for i in range(B)]
    dit.set_precision(precision)
    try:
        a = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), seed=seeds, **kw)["target_latents"]
        b = generate_latents(dit, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), seed=seeds, **kw)["target_latents"]
        solo = generate_latents(dit, null, enc, ctx1, seed=[seeds[1]], **kw)["target_latents"]
    finally:
        dit.set_precision("bf16")
    assert torch.isfinite(a).all() and float(a.std()) > 0.1
    assert torch.equal(a, b), "same request twice must be bit-identical"
    r = _rel(a[1:2], solo)
    print(f"{name} ({precision}): item 1 of {B} vs alone, rel L2 {r:.3e}")
    assert r < (2e-2 if precision == "mxfp8" else 5e-3), r   # MXFP8: the quantiser sees the same rows either way; only tile order differs
    assert _rel(a[0:1], a[1:2]) > 0.5
