"""The tolerance SURVEY.md section 8(d) states for the bf16 path: "threshold = 2 x the drift the CPU restatement shows between fp32 and
bf16-weights / fp32-accumulate runs of itself".  This script measures that drift for every golden case of the DiT path and writes it
to tests/golden/bf16_storage_drift.json; the GPU tests gate the HIP path at twice the recorded figure (tests/_drift.py).

What is run: the oracle (oracle/dit.py, oracle/sampler.py - fp32 arithmetic) on the fixture's own inputs in its bf16-STORAGE mode
(`oracle.dit.bf16_storage()`: every matrix of the weight set and every operand of a contraction - linear / conv inputs, Q, K, V, the softmax
probabilities - rounded to bfloat16 on its way in; accumulation, norms, residual stream, guidance and the integrator stay fp32: the
storage precision of the reference's own GPU path, handler/init_service_orchestrator.py:51, and of the HIP path), compared with the
fixture's expected output - which IS the fp32 run (the reference's, equal to the oracle's
to <= 2e-6: make_golden.py).  Needs no reference checkout: inputs are rebuilt from the seeds the fixtures record and pinned by their
checksums.  The full-size cases (G11-G16) cost ~ 800 TFLOP of fp32 CPU work (about 45 minutes on 8 cores); the tiny ones a few seconds -
tests/test_drift_table.py recomputes those on every CPU run so that the table cannot rot unnoticed.

    python tests/golden/make_drift.py            # tiny cases only (merged into the existing table)
    python tests/golden/make_drift.py --full     # everything
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TINY = dict(hidden_size=256, intermediate_size=768, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
TABLE = os.path.join(HERE, "bf16_storage_drift.json")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


class _Keep(dict):
    """A taps dict that keeps only the named entries (the oracle offers every intermediate of every layer: 8 GB at N = 16)."""

    def __init__(self, names):
        super().__init__()
        self._names = set(names)

    def __setitem__(self, k, v):
        if k in self._names:
            super().__setitem__(k, v)


def tiny_cases():
    import ace355
    from ace355 import weightgen
    from oracle import dit as o_dit, sampler as o_s
    from oracle.dit import bf16_weights
    out = {}
    G = np.load(os.path.join(HERE, "g2_tiny_forward.npz"))
    for case in "abc":
        window = int(G[f"{case}_window"])
        cfg = ace355.DitConfig(**TINY, sliding_window=window)
        wb = bf16_weights(weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=int(G["seed"]), mode="test"))
        x, ctx, enc, t = (torch.from_numpy(G[f"{case}_{k}"]) for k in ("x", "ctx", "enc", "t"))
        v = o_dit.dit_forward(o_dit.DitConfig(**TINY, sliding_window=window), wb, x, t, t, enc, ctx)
        out[f"g2/{case}"] = {"v": rel(v, torch.from_numpy(G[f"{case}_v"]))}
    G = np.load(os.path.join(HERE, "g3_tiny_sampler.npz"))
    cfg = ace355.DitConfig(**TINY)
    wb = bf16_weights(weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=int(G["seed"]), mode="test"))
    null = weightgen.make_null_condition_emb(cfg.hidden_size, seed=int(G["seed"]))
    for name in ("cfg7_shift1", "cfg1_shift3", "cfg7_interval", "sft_timesteps"):
        enc, ctx = torch.from_numpy(G[f"{name}_enc"]), torch.from_numpy(G[f"{name}_ctx"])
        B = ctx.shape[0]
        lo, hi = G[f"{name}_interval"].tolist()
        res = o_s.generate_audio(o_dit.DitConfig(**TINY), wb, null, enc.expand(B, -1, -1), ctx, seed=G[f"{name}_seeds"].tolist(),
                                 infer_steps=int(G[f"{name}_steps"]), diffusion_guidance_sale=float(G[f"{name}_guidance"]),
                                 cfg_interval_start=lo, cfg_interval_end=hi, shift=float(G[f"{name}_shift"]),
                                 timesteps=G[f"{name}_timesteps"].tolist() or None)
        out[f"g3/{name}"] = {"out": rel(res, torch.from_numpy(G[f"{name}_out"]))}
    return out


def metric_inputs(B, T, seed0=1000, ctx_seed=45):
    """= tests/test_metric_shapes_gpu.py::_inputs = make_golden.metric_inputs"""
    from oracle.sampler import prepare_noise
    x = prepare_noise((B, T, 64), [seed0 + i for i in range(B)])
    g = torch.Generator().manual_seed(ctx_seed)
    ctx1 = torch.cat([0.5 * torch.randn(1, T, 64, generator=g), torch.ones(1, T, 64)], -1)
    return x, ctx1


def _close(a, b):
    return abs(a - b) <= 1e-9 * abs(b)


def full_cases(only=None):
    import ace355
    from ace355 import weightgen
    from oracle import dit as o_dit, sampler as o_s
    from oracle.dit import bf16_weights
    cfg = ace355.DitConfig()
    o_cfg = o_dit.DitConfig()
    t0 = time.time()
    wb = bf16_weights(weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=4, mode="test"))
    null = weightgen.make_null_condition_emb(cfg.hidden_size, seed=4)
    enc = torch.from_numpy(np.load(os.path.join(HERE, "g4_full_forward.npz"))["enc"])
    print(f"weights ready ({time.time() - t0:.0f} s)", flush=True)
    out = {}

    def want(name):
        return not only or name in only

    def sampler(G, B, T=750):
        _, ctx1 = metric_inputs(B, T)
        assert _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
        res = o_s.generate_audio(o_cfg, wb, null, enc.expand(B, -1, -1), ctx1.expand(B, -1, -1).contiguous(), seed=G["seeds"].tolist(),
                                 infer_steps=int(G["steps"]), diffusion_guidance_sale=float(G["guidance"]))
        ref = torch.from_numpy(G["out"])
        d2 = [float((res[i].double() - ref[i].double()).pow(2).sum()) for i in range(B)]
        r2 = [float(ref[i].double().pow(2).sum()) for i in range(B)]
        # per-song squared norms: the drift of ANY slice of the request follows (tests/_drift.py: slice_drift)
        return {"out": rel(res, ref), "per_song_max": max((d / r) ** 0.5 for d, r in zip(d2, r2)), "diff2": d2, "ref2": r2}

    if want("g4"):
        G = np.load(os.path.join(HERE, "g4_full_forward.npz"))
        x, ctx, e1, t = (torch.from_numpy(G[k]) for k in ("x", "ctx", "enc", "t"))
        e = torch.cat([e1[:1], null.reshape(1, 1, -1).expand(1, e1.shape[1], -1)], 0)
        taps = _Keep(["l0.out", "l23.out"])
        v = o_dit.dit_forward(o_cfg, wb, x, t, t, e, ctx, taps=taps)
        out["g4"] = {"v": rel(v, torch.from_numpy(G["v"])), "l0": rel(taps["l0.out"][:, ::25], torch.from_numpy(G["l0_out"])),
                     "l23": rel(taps["l23.out"][:, ::25], torch.from_numpy(G["l23_out"]))}
        print("g4", out["g4"], f"({time.time() - t0:.0f} s)", flush=True)
    if want("g11"):
        G = np.load(os.path.join(HERE, "g11_metric_forward.npz"))
        B, T = 8, 750
        x8, ctx1 = metric_inputs(B, T)
        assert _close(float(x8.double().abs().sum()), float(G["x_sum"])) and _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
        x = torch.cat([x8, x8], 0)
        e = torch.cat([enc[:1].expand(B, -1, -1), null.reshape(1, 1, -1).expand(B, enc.shape[1], -1)], 0)
        t = torch.full((2 * B,), float(G["t"]))
        taps = _Keep(["l0.out", "l23.out"])
        v = o_dit.dit_forward(o_cfg, wb, x, t, t, e, ctx1.expand(2 * B, -1, -1).contiguous(), taps=taps)
        ref = torch.from_numpy(G["v"])
        seqs, stride = G["tap_seqs"].tolist(), int(G["tap_stride"])
        out["g11"] = {"v": rel(v, ref), "v_cond": rel(v[:B], ref[:B]), "v_null": rel(v[B:], ref[B:]),
                      "l0": rel(taps["l0.out"][seqs][:, ::stride], torch.from_numpy(G["l0_out"])),
                      "l23": rel(taps["l23.out"][seqs][:, ::stride], torch.from_numpy(G["l23_out"]))}
        print("g11", out["g11"], f"({time.time() - t0:.0f} s)", flush=True)
    for name, fn, T, seed0, ctx_seed in (("g13", "g13_120s_forward.npz", 3000, 2000, 46), ("g14", "g14_240s_forward.npz", 6000, 4000, 47)):
        if not want(name):
            continue
        G = np.load(os.path.join(HERE, fn))
        x1, ctx1 = metric_inputs(1, T, seed0=seed0, ctx_seed=ctx_seed)
        assert _close(float(x1.double().abs().sum()), float(G["x_sum"])) and _close(float(ctx1.double().abs().sum()), float(G["ctx_sum"]))
        e = torch.cat([enc[:1], null.reshape(1, 1, -1).expand(1, enc.shape[1], -1)], 0)
        t = torch.full((2,), float(G["t"]))
        taps = _Keep(["l23.out"])
        v = o_dit.dit_forward(o_cfg, wb, torch.cat([x1, x1], 0), t, t, e, ctx1.expand(2, -1, -1).contiguous(), taps=taps)
        out[name] = {"v": rel(v, torch.from_numpy(G["v"]))}
        if "l23_out" in G.files:
            out[name]["l23"] = rel(taps["l23.out"][:, ::100], torch.from_numpy(G["l23_out"]))
        print(name, out[name], f"({time.time() - t0:.0f} s)", flush=True)
    for name, fn, B in (("g12", "g12_metric_sampler.npz", 8), ("g15", "g15_full_schedule_sampler.npz", 3), ("g16", "g16_bench_request_sampler.npz", 8)):
        if want(name):
            out[name] = sampler(np.load(os.path.join(HERE, fn)), B)
            print(name, {k: v_ for k, v_ in out[name].items() if k not in ("diff2", "ref2")}, f"({time.time() - t0:.0f} s)", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.manual_seed(0)
    table = json.load(open(TABLE)) if os.path.exists(TABLE) else {}
    table["_what"] = ("rel L2 between the oracle in its bf16-storage mode (weights and contraction operands rounded to bfloat16, fp32 arithmetic) and the fixture's fp32 expectation, per golden case; "
                      "the GPU tests gate the HIP path at 2 x these (SURVEY.md 8d); made by tests/golden/make_drift.py")
    from oracle.dit import bf16_storage
    with bf16_storage():
        table.update(tiny_cases())
        if args.full:
            table.update(full_cases(set(args.only.split(",")) - {""} or None))
    json.dump(table, open(TABLE, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: ({a: b for a, b in v.items() if a not in ("diff2", "ref2")} if isinstance(v, dict) else v) for k, v in table.items() if k != "_what"}, indent=1))


if __name__ == "__main__":
    main()
