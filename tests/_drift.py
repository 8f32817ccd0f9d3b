"""The tolerance of the bf16 HIP path, as SURVEY.md section 8(d) states it: "threshold = 2 x the drift the CPU restatement shows between
fp32 and bf16 runs of itself (stated, measured in the same run)".

The restatement's bf16 run is `oracle.dit.bf16_storage()`: weight matrices and every operand of a contraction (linear / conv inputs, Q, K, V,
softmax probabilities) rounded to bfloat16 on their way in, everything else - accumulation, norms, residual stream, guidance, integrator -
fp32.  That is the storage precision of the reference's own GPU path (handler/init_service_orchestrator.py:51) and of the HIP kernels.

* `emulated(fn, cfg, w, ...)` runs an oracle function that way; tests whose fp32 expectation comes from the oracle (or from a tiny golden
  fixture) measure the drift in the same run and gate the HIP result at twice it.
* For the full-size fixtures (G11-G16: minutes to hours of fp32 CPU work per case) the drift was measured once by tests/golden/make_drift.py
  and is read from tests/golden/bf16_storage_drift.json (`table`); tests/test_drift_table.py recomputes the table's tiny entries on every CPU
  run, so the script and the table cannot drift apart unnoticed.
* `check` prints measured value, drift and gate, asserts measured < FACTOR x drift, and returns the ratio.
"""
import json
import os

FACTOR = 2.0
_TABLE = None


def table(case, key=None):
    global _TABLE
    if _TABLE is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_storage_drift.json")) as f:
            _TABLE = json.load(f)
    return _TABLE[case] if key is None else _TABLE[case][key]


def slice_drift(case, n):
    """Drift of the first n songs of a recorded request (per-song squared norms are in the table)."""
    e = table(case)
    return (sum(e["diff2"][:n]) / sum(e["ref2"][:n])) ** 0.5


def emulated(fn, cfg, w, *args, **kw):
    """fn(cfg, bf16-rounded w, *args) with the oracle in its bf16-storage mode."""
    from oracle import dit as o_dit
    with o_dit.bf16_storage():
        return fn(cfg, o_dit.bf16_weights(w), *args, **kw)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def check(what, measured, drift, factor=FACTOR):
    gate = factor * drift
    print(f"    [8d gate] {what}: measured {measured:.3e}, oracle's own bf16-storage drift {drift:.3e} -> gate {gate:.3e} "
          f"(measured / drift = {measured / drift:.2f})")
    assert measured < gate, f"{what}: {measured:.3e} >= {factor} x {drift:.3e} (the oracle's bf16-storage drift)"
    return measured / drift
