"""CPU: the table behind the GPU parity gates (tests/golden/bf16_storage_drift.json, SURVEY.md section 8d: gate = 2 x the oracle's own
bf16-storage drift) is what tests/golden/make_drift.py produces - its tiny entries are recomputed here on every run - and the oracle's
bf16-storage mode leaves the default fp32 path alone."""
import importlib.util
import os

import numpy as np
import torch

import _drift

HERE = os.path.dirname(os.path.abspath(__file__))


def _make_drift():
    spec = importlib.util.spec_from_file_location("make_drift", os.path.join(HERE, "golden", "make_drift.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_tiny_entries_of_the_table_are_reproduced_and_the_full_size_ones_are_there():
    from oracle import dit as o_dit
    md = _make_drift()
    with o_dit.bf16_storage():
        now = md.tiny_cases()
    assert set(now) == {"g2/a", "g2/b", "g2/c", "g3/cfg7_shift1", "g3/cfg1_shift3", "g3/cfg7_interval", "g3/sft_timesteps"}
    for case, vals in now.items():
        for key, v in vals.items():
            t = _drift.table(case, key)
            assert abs(v / t - 1) < 0.02, (case, key, v, t)   # (fp32 summation order differs with the host's thread count: a few bf16 roundings flip)
    # full-size fixtures: every key the GPU tests read, in the range bf16 storage of a 24-layer model can produce
    want = {"g4": ("v", "l0", "l23"), "g11": ("v", "v_cond", "v_null", "l0", "l23"), "g12": ("out", "per_song_max"), "g13": ("v", "l23"), "g14": ("v",),
            "g15": ("out", "per_song_max"), "g16": ("out", "per_song_max", "diff2", "ref2")}
    for case, keys in want.items():
        for key in keys:
            v = _drift.table(case, key)
            if key in ("diff2", "ref2"):
                assert len(v) == 8 and all(x > 0 for x in v)
            else:
                assert 1e-3 < v < 1e-2, (case, key, v)
    assert abs(_drift.slice_drift("g16", 8) / _drift.table("g16", "out") - 1) < 1e-9
    assert 1e-3 < _drift.slice_drift("g16", 1) < 1e-2


def test_bf16_storage_mode_is_scoped_and_rounds_operands_only():
    from oracle import dit as o_dit
    G = np.load(os.path.join(HERE, "golden", "g2_tiny_forward.npz"))
    md = _make_drift()
    import ace355
    from ace355 import weightgen
    window = int(G["a_window"])
    cfg = ace355.DitConfig(**md.TINY, sliding_window=window)
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=int(G["seed"]), mode="test")
    x, ctx, enc, t = (torch.from_numpy(G[f"a_{k}"]) for k in ("x", "ctx", "enc", "t"))
    o_cfg = o_dit.DitConfig(**md.TINY, sliding_window=window)
    ref = torch.from_numpy(G["a_v"])
    assert _drift.rel(o_dit.dit_forward(o_cfg, w, x, t, t, enc, ctx), ref) < 1e-5
    emu = _drift.emulated(o_dit.dit_forward, o_cfg, w, x, t, t, enc, ctx)
    assert not o_dit._BF16_STORAGE                                                      # the context is left
    assert _drift.rel(o_dit.dit_forward(o_cfg, w, x, t, t, enc, ctx), ref) < 1e-5      # ... and the default path is the fp32 one again
    d_all = _drift.rel(emu, ref)
    d_w = _drift.rel(o_dit.dit_forward(o_cfg, o_dit.bf16_weights(w), x, t, t, enc, ctx), ref)   # weights only
    assert 1e-3 < d_w < d_all < 1e-2, (d_w, d_all)
    wb = o_dit.bf16_weights(w)
    assert all(torch.equal(wb[k], w[k]) for k in w if w[k].ndim < 2 or "scale_shift" in k)
    assert any(not torch.equal(wb[k], w[k]) for k in w if w[k].ndim >= 2)
    assert _drift.check("self-test", 1.0e-3, 0.6e-3) > 1.6
    try:
        _drift.check("self-test", 1.3e-3, 0.6e-3)
    except AssertionError:
        pass
    else:
        raise AssertionError("a value above twice the drift passed")
