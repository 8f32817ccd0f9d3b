"""GPU parity of the audio-token detokenizer through the C ABI, against golden vectors captured
from the imported reference (tests/golden/g8_detokenizer.npz) and against the oracle at the real architecture."""
import numpy as np
import pytest
import torch

import _drift

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def test_detokenizer_vs_reference_golden(gpu_device, golden_dir):
    import ace355
    from ace355 import weightgen
    from ace355.lmhints import NativeDetokenizer
    G = np.load(f"{golden_dir}/g8_detokenizer.npz")
    cfg = ace355.DetokConfig(hidden_size=256, intermediate_size=768, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=int(G["seed"]), mode="test")
    assert abs(weightgen.checksum(w) - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    det = NativeDetokenizer(cfg, gpu_device)
    det.load_state_dict(w)
    y = det(torch.from_numpy(G["x"]))
    ref = torch.from_numpy(G["y"])
    assert y.shape == ref.shape
    r = _rel(y.cpu(), ref)
    from oracle import detok as o_detok
    o_cfg = o_detok.DetokConfig(hidden_size=256, intermediate_size=768, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    emu = _drift.emulated(o_detok.detokenizer, o_cfg, w, torch.from_numpy(G["x"]))
    print(f"detokenizer (tiny) vs reference fp32: rel L2 {r:.3e}; vs the oracle with bf16 storage {_rel(y.cpu(), emu):.3e}")
    _drift.check("detokenizer (tiny)", r, _rel(emu, ref))   # reference fp32 CPU vs bf16 kernels, 2 layers; measured 3.66e-3 against a drift of 3.67e-3


def test_detokenizer_full_size_vs_oracle_and_code_path(gpu_device):
    """Real architecture (2 layers, 2048 wide) on a 10 s clip's worth of codes (50 tokens -> 250 frames), fed through the
    whole N2 chain: code string -> indices -> FSQ output -> detokenizer."""
    import ace355
    from ace355 import weightgen
    from ace355.lmhints import NativeDetokenizer, decode_audio_codes_to_latents, parse_audio_code_string
    from oracle import detok as o_detok
    cfg = ace355.DetokConfig()
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=12, mode="test")
    det = NativeDetokenizer(cfg, gpu_device)
    det.load_state_dict(w)
    g = torch.Generator().manual_seed(6)
    codes = torch.randint(0, 64000, (50,), generator=g).tolist()
    s = "".join(f"<|audio_code_{c}|>" for c in codes)
    assert parse_audio_code_string(s) == codes == o_detok.parse_audio_code_string(s)
    pw = 0.5 * torch.randn(cfg.hidden_size, 6, generator=g)
    pb = 0.1 * torch.randn(cfg.hidden_size, generator=g)
    y = decode_audio_codes_to_latents(s, det, pw.to(gpu_device), pb.to(gpu_device))
    q = o_detok.fsq_output_from_indices(torch.tensor(codes).view(1, -1, 1), o_detok.FSQ_LEVELS if hasattr(o_detok, "FSQ_LEVELS") else (8, 8, 8, 5, 5, 5), pw, pb)
    ref = o_detok.detokenizer(o_detok.DetokConfig(), w, q)
    assert y.shape == (1, 250, 64) == tuple(ref.shape)
    r = _rel(y.cpu(), ref)
    emu = _drift.emulated(o_detok.detokenizer, o_detok.DetokConfig(), w, q)
    print(f"detokenizer (full size) vs fp32 oracle: rel L2 {r:.3e}; vs the oracle with bf16 storage {_rel(y.cpu(), emu):.3e}")
    _drift.check("detokenizer (full size)", r, _rel(emu, ref))   # measured 5.75e-3
    assert torch.equal(y, decode_audio_codes_to_latents(s, det, pw.to(gpu_device), pb.to(gpu_device))), "the same codes twice: not bit-identical"
    assert decode_audio_codes_to_latents("no codes here", det, pw, pb) is None
    with pytest.raises(ValueError):
        det(torch.zeros(1, 4, cfg.hidden_size), attention_mask=torch.ones(1, 4))


def test_audio_tokenizer_vs_reference_golden(gpu_device, golden_dir):
    """G17: AceStepAudioTokenizer up to its quantizer (audio_acoustic_proj + AttentionPooler, base.py:734-859, 1181-1218) through
    ace355_tok_* against the vector captured from the imported reference."""
    import ace355
    from ace355 import weightgen
    from ace355.lmhints import NativeAudioTokenizer
    from oracle import detok as o_detok
    G = np.load(f"{golden_dir}/g17_audio_tokenizer.npz")
    cfg = ace355.DetokConfig(hidden_size=256, intermediate_size=768, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    o_cfg = o_detok.DetokConfig(hidden_size=256, intermediate_size=768, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    w = weightgen.make_dit_weights(o_detok.tok_weight_shapes(o_cfg), cfg.hidden_size, seed=int(G["seed"]), mode="test")
    assert abs(weightgen.checksum(w) - float(G["wsum"])) < 1e-6 * float(G["wsum"])
    tok = NativeAudioTokenizer(cfg, gpu_device)
    tok.load_state_dict(w)
    y = tok.pool(torch.from_numpy(G["x"]))
    ref = torch.from_numpy(G["y"])
    assert y.shape == ref.shape
    r = _rel(y.cpu(), ref)
    x = torch.from_numpy(G["x"])
    P = o_cfg.pool_window_size
    emu = _drift.emulated(o_detok.tokenizer_pool, o_cfg, w, x.reshape(x.shape[0], x.shape[1] // P, P, x.shape[2]))
    print(f"audio tokenizer (tiny; projection + attention pooler) vs reference fp32: rel L2 {r:.3e}; vs the oracle with bf16 storage {_rel(y.cpu(), emu):.3e}")
    _drift.check("audio tokenizer (tiny)", r, _rel(emu, ref))   # measured 6.9e-3
    with pytest.raises(RuntimeError, match="quantizer"):
        tok(torch.from_numpy(G["x"]))   # no quantizer.* weights were loaded
    with pytest.raises(ValueError):
        tok.pool(torch.zeros(1, 7, 64))  # not whole windows


def test_audio_tokenizer_full_size_round_trip_through_the_detokenizer(gpu_device):
    """Real architecture: 10 s of latents -> NativeAudioTokenizer (pooler native, FSQ restated) -> NativeDetokenizer, the chain
    prepare_condition runs for a cover task without precomputed hints (base.py:1645-1649), against the oracle's chain."""
    import ace355
    from ace355 import weightgen
    from ace355.lmhints import NativeAudioTokenizer, NativeDetokenizer, fsq_output_from_indices
    from oracle import detok as o_detok
    cfg = ace355.DetokConfig()
    o_cfg = o_detok.DetokConfig()
    wt = weightgen.make_dit_weights(o_detok.tok_weight_shapes(o_cfg), cfg.hidden_size, seed=21, mode="test")
    wd = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=22, mode="test")
    g = torch.Generator().manual_seed(9)
    q = {"quantizer.project_in.weight": 0.05 * torch.randn(6, cfg.hidden_size, generator=g), "quantizer.project_in.bias": 0.1 * torch.randn(6, generator=g),
         "quantizer.project_out.weight": 0.5 * torch.randn(cfg.hidden_size, 6, generator=g), "quantizer.project_out.bias": 0.1 * torch.randn(cfg.hidden_size, generator=g)}
    tok = NativeAudioTokenizer(cfg, gpu_device)
    tok.load_state_dict({**wt, **q})
    det = NativeDetokenizer(cfg, gpu_device)
    det.load_state_dict(wd)
    x = torch.randn(2, 250, 64, generator=g)
    pooled = tok.pool(x)
    ref_pool = o_detok.tokenizer_pool(o_cfg, wt, x.reshape(2, 50, 5, 64))
    rp = _rel(pooled.cpu(), ref_pool)
    quant, idx = tok.tokenize(x)
    assert torch.equal(pooled, tok.pool(x)), "the same frames twice: not bit-identical"
    assert quant.shape == (2, 50, cfg.hidden_size) and idx.shape == (2, 50, 1) and int(idx.min()) >= 0 and int(idx.max()) < 64000
    # the two FSQ restatements agree with each other: decoding the indices gives the quantised output back
    back = fsq_output_from_indices(idx, q["quantizer.project_out.weight"].to(gpu_device), q["quantizer.project_out.bias"].to(gpu_device))
    assert float((back - quant).abs().max()) < 1e-5
    # the same rounding decisions as the oracle's quantizer wherever the pooled value is not within bf16 noise of a rounding boundary
    oq, oidx = o_detok.fsq_quantize(ref_pool, o_detok.FSQ_LEVELS if hasattr(o_detok, "FSQ_LEVELS") else (8, 8, 8, 5, 5, 5), q["quantizer.project_in.weight"],
                                    q["quantizer.project_in.bias"], q["quantizer.project_out.weight"], q["quantizer.project_out.bias"])
    agree = float((oidx == idx.cpu()).float().mean())
    hints = det(quant)
    assert hints.shape == (2, 250, 64) and bool(torch.isfinite(hints).all()) and torch.equal(hints, det(quant))
    print(f"audio tokenizer (full size): pooled rel L2 vs fp32 oracle {rp:.3e}; {100 * agree:.0f} % of the 100 tokens get the oracle's code index")
    emu_pool = _drift.emulated(o_detok.tokenizer_pool, o_cfg, wt, x.reshape(2, 50, 5, 64))
    _drift.check("audio tokenizer (full size), pooled", rp, _rel(emu_pool, ref_pool))   # measured 1.07e-2 (two plain-residual layers + one more bf16 operand than the detokenizer)
    assert agree > 0.7, agree     # measured 0.93


@pytest.mark.parametrize("M,N,K", [(1200, 6, 2048), (1200, 2048, 6), (3, 5, 65), (1, 1, 1)])
def test_fsq_projection_kernel_vs_fp64(gpu_device, M, N, K):
    """``ace355_linear_f32`` (the residual FSQ's project_in / project_out on the LM-hint path, H/audio_codes.py:47-66): fp32 in / out,
    fp64 accumulation, against torch fp64 on the host; both kernel forms (one lane / one wave per output); same bits twice."""
    from ace355.lmhints import _linear
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(4, M // 4 if M % 4 == 0 else M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    ref = (x.double() @ w.double().t() + b.double()).float()
    y = _linear(x, w.to(gpu_device), b.to(gpu_device))
    y2 = _linear(x, w.to(gpu_device), b.to(gpu_device))
    assert y.shape == ref.shape and torch.equal(y, y2)
    assert float((y.cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()) + 1e-7   # one fp32 rounding of the fp64 sum
    with pytest.raises(RuntimeError, match="native library"):
        _linear(x, w, b)   # CPU weights: no fallback
