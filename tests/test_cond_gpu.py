"""GPU parity of the condition encoder through the C ABI, against the oracle and against the
golden vectors captured from the imported reference (tests/golden/g7_cond_encoder.npz).

bf16 storage / fp32 accumulate vs the fp32 reference: tolerances are relative L2 over ALL rows of the packed output,
padding rows included (the DiT attends them, base.py:1384-1385).
"""
import ctypes as C

import numpy as np
import pytest
import torch

import _drift

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def T(x):
    return torch.from_numpy(np.asarray(x))


@pytest.mark.parametrize("N,S,H,Hkv,window,lens", [
    (3, 300, 2, 1, 128, [300, 90, 1]),      # item 1: queries > 218 have no valid key on the band -> uniform over all keys
    (2, 200, 4, 2, 16, [37, 0]),            # a sequence with no valid key at all
    (2, 129, 2, 2, -1, [129, 64]),          # full attention: padded keys only
])
def test_attention_key_padding_mask(gpu_device, N, S, H, Hkv, window, lens):
    from ace355 import native
    from oracle import cond as o_cond
    from oracle import dit as o_dit
    lib = native.lib()
    g = torch.Generator().manual_seed(S + H)
    bf = lambda x: x.to(torch.bfloat16)  # noqa: E731
    q = bf(torch.randn(N, S, H * 128, generator=g))
    k = bf(torch.randn(N, S, Hkv * 128, generator=g))
    v = bf(torch.randn(N, S, Hkv * 128, generator=g) + torch.arange(S)[None, :, None] * 0.01)
    am = (torch.arange(S)[None, :] < torch.tensor(lens)[:, None]).long()
    mask = o_cond.mask_4d(S, am, window if window >= 0 else None)
    ref = o_dit.attention(q.float().view(N, S, H, 128).transpose(1, 2), k.float().view(N, S, Hkv, 128).transpose(1, 2),
                          v.float().view(N, S, Hkv, 128).transpose(1, 2), mask, 128 ** -0.5)
    out = torch.empty(N, S, H * 128, device=gpu_device, dtype=torch.bfloat16)
    qd, kd, vd = q.to(gpu_device), k.to(gpu_device), v.to(gpu_device)
    kv = (C.c_int32 * N)(*lens)
    native.check(lib.ace355_attention_masked(native.ptr(qd), native.ptr(kd), native.ptr(vd), native.ptr(out), N, S, S, H, Hkv, window,
                                             128 ** -0.5, kv, None), "attention_masked")
    torch.cuda.synchronize()
    assert _rel(out.cpu(), ref) < 1e-2, _rel(out.cpu(), ref)
    # rows with no valid key: mean of V over all keys (checked separately, they are a small part of the norm above)
    if window >= 0:
        b = 1 if lens[1] < S else 0
        far = min(S - 1, lens[b] + window + 5)
        assert _rel(out.cpu()[b, far], ref[b, far]) < 1e-2


def _tiny_cfg(window):
    import ace355
    return ace355.CondConfig(hidden_size=256, intermediate_size=768, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                             text_hidden_dim=64, timbre_hidden_dim=64, num_lyric_encoder_hidden_layers=2,
                             num_timbre_encoder_hidden_layers=2, sliding_window=window)


@pytest.mark.parametrize("case", ["a", "b"])
def test_condition_encoder_vs_reference_golden(gpu_device, golden_dir, case):
    from ace355 import weightgen
    from ace355.cond import NativeCondEncoder
    G = np.load(f"{golden_dir}/g7_cond_encoder.npz")
    cfg = _tiny_cfg(int(G[f"{case}_window"]))
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=int(G["seed"]), mode="test")
    assert abs(weightgen.checksum(w) - float(G[f"{case}_wsum"])) < 1e-6 * float(G[f"{case}_wsum"])
    enc = NativeCondEncoder(cfg, gpu_device)
    enc.load_state_dict(w)
    h, m = enc(T(G[f"{case}_text"]), T(G[f"{case}_tmask"]), T(G[f"{case}_lyric"]), T(G[f"{case}_lmask"]), T(G[f"{case}_refer"]),
               T(G[f"{case}_order"]))
    ref_h, ref_m = T(G[f"{case}_h"]), T(G[f"{case}_m"])
    assert h.shape == ref_h.shape and torch.equal(m.cpu().long(), ref_m)
    # reference fp32 CPU vs bf16 kernels, 2 + 2 layers: 2e-2 relative L2 over all rows; valid rows alone as well
    vm = ref_m.bool()
    ra, rv, rp = _rel(h.cpu(), ref_h), _rel(h.cpu()[vm], ref_h[vm]), _rel(h.cpu()[~vm], ref_h[~vm])
    print(f"condition encoder case {case}: rel L2 vs reference fp32 all rows {ra:.3e}, valid {rv:.3e}, padding {rp:.3e}")
    # gates = 2 x the oracle's own bf16-storage drift on this fixture, per row class (measured 3.3e-3 - 3.8e-3; padding rows incl. uniform-attention rows
    # and the zero timbre rows)
    from oracle import cond as o_cond
    o_cfg = o_cond.CondConfig(**{k: getattr(cfg, k) for k in ("hidden_size", "intermediate_size", "num_attention_heads", "num_key_value_heads", "head_dim",
                                                              "text_hidden_dim", "timbre_hidden_dim", "num_lyric_encoder_hidden_layers",
                                                              "num_timbre_encoder_hidden_layers", "sliding_window")})
    emu, _m = _drift.emulated(o_cond.condition_encoder, o_cfg, w, T(G[f"{case}_text"]), T(G[f"{case}_tmask"]), T(G[f"{case}_lyric"]), T(G[f"{case}_lmask"]),
                              T(G[f"{case}_refer"]), T(G[f"{case}_order"]))
    print(f"    vs the oracle with bf16 storage: all rows {_rel(h.cpu(), emu):.3e}")
    _drift.check(f"condition encoder {case}, all rows", ra, _rel(emu, ref_h))
    _drift.check(f"condition encoder {case}, valid rows", rv, _rel(emu[vm], ref_h[vm]))
    _drift.check(f"condition encoder {case}, padding rows", rp, _rel(emu[~vm], ref_h[~vm]))


def test_condition_encoder_full_size_vs_oracle(gpu_device):
    """Real architecture (8 + 4 layers, 2048 wide, 608 M parameters), short sequences so the fp32 oracle stays in seconds."""
    import ace355
    from ace355 import weightgen
    from ace355.cond import NativeCondEncoder
    from oracle import cond as o_cond
    cfg = ace355.CondConfig()
    w = weightgen.make_dit_weights(cfg.weight_shapes(), cfg.hidden_size, seed=11, mode="test")
    o_cfg = o_cond.CondConfig()
    B, Lt, Ll, Tref = 2, 24, 200, 96
    g = torch.Generator().manual_seed(5)
    text = torch.randn(B, Lt, cfg.text_hidden_dim, generator=g)
    lyric = torch.randn(B, Ll, cfg.text_hidden_dim, generator=g)
    refer = torch.randn(3, Tref, cfg.timbre_hidden_dim, generator=g)
    tmask = (torch.arange(Lt)[None, :] < torch.tensor([24, 9])[:, None]).long()
    lmask = (torch.arange(Ll)[None, :] < torch.tensor([200, 41])[:, None]).long()
    order = torch.tensor([0, 0, 1])
    ref_h, ref_m = o_cond.condition_encoder(o_cfg, w, text, tmask, lyric, lmask, refer, order)
    enc = NativeCondEncoder(cfg, gpu_device)
    enc.load_state_dict(w)
    h, m = enc(text, tmask, lyric, lmask, refer, order)
    assert torch.equal(m.cpu(), ref_m)
    r = _rel(h.cpu(), ref_h)
    # Reproducibility (this value wandered between 1.29e-2 and 1.59e-2 for five evidence runs under a 3e-2 gate, the output
    # changed from call to call; cause and fix: DESIGN.md section 14): four more calls must return the SAME bits, and the bits are the ones
    # recorded here when the fault was fixed (sha256 of the fp32 output; a deliberate change of the encoder's arithmetic updates it).
    import hashlib
    sha = hashlib.sha256(h.cpu().numpy().tobytes()).hexdigest()[:16]
    for _ in range(4):
        h2, _m = enc(text, tmask, lyric, lmask, refer, order)
        assert torch.equal(h, h2), f"the same request twice: {_rel(h2.cpu(), h.cpu()):.3e} apart"
    print(f"condition encoder full size: rel L2 vs fp32 oracle {r:.3e}, output sha {sha}, 5 calls bit-identical")
    # measured 1.166e-2 in rounds 2-5 before the fault and again once it was fixed (sha d8528bec82a8c1a6); 1.169e-2 since the head-norm epilogue sums a
    # row's squares block by block in every tile form (another fp32 order of the same sum): 12 bf16 layers with PLAIN residuals
    emu, _m = _drift.emulated(o_cond.condition_encoder, o_cfg, w, text, tmask, lyric, lmask, refer, order)
    print(f"    vs the oracle with bf16 storage: {_rel(h.cpu(), emu):.3e}")
    _drift.check("condition encoder, full size", r, _rel(emu, ref_h))
    assert sha == "14079fa2dede1709", sha
    # and its output drives the DiT's condition slot unchanged
    assert h.dtype == torch.float32 and h.is_contiguous() and h.shape == (B, Ll + 2 + Lt, cfg.hidden_size)


def test_condition_encoder_rejects_non_prefix_masks(gpu_device):
    from ace355.cond import NativeCondEncoder
    cfg = _tiny_cfg(16)
    enc = NativeCondEncoder(cfg, gpu_device)
    text, lyric, refer = torch.zeros(1, 4, 64), torch.zeros(1, 8, 64), torch.zeros(1, 8, 64)
    with pytest.raises(ValueError, match="prefix mask"):
        enc(text, torch.tensor([[1, 0, 1, 0]]), lyric, torch.ones(1, 8, dtype=torch.long), refer, torch.tensor([0]))
    with pytest.raises(RuntimeError, match="finalize"):
        enc(text, torch.ones(1, 4, dtype=torch.long), lyric, torch.ones(1, 8, dtype=torch.long), refer, torch.tensor([0]))


def test_request_chain_cond_encoder_to_sampler_to_vae(gpu_device):
    """The widened path end to end on one request (tiny widths, real depth pattern): condition encoder -> condition slots ->
    27-step... here 6-step CFG+APG sampler -> VAE decode, native vs the oracle chain on the same inputs.  Every hand-over is
    the product's own: the encoder's packed output (padding rows included) is what the DiT attends."""
    import ace355
    from ace355 import weightgen
    from ace355.cond import NativeCondEncoder
    from ace355.dit import NativeDit, generate_latents
    from ace355.vae import NativeVae
    from oracle import cond as o_cond, dit as o_dit, oobleck as o_vae, sampler as o_sampler
    tiny = dict(hidden_size=256, intermediate_size=768, num_attention_heads=2, num_key_value_heads=1, head_dim=128)
    ccfg = ace355.CondConfig(**tiny, text_hidden_dim=64, timbre_hidden_dim=64, num_lyric_encoder_hidden_layers=2, num_timbre_encoder_hidden_layers=2)
    dcfg = ace355.DitConfig(**tiny, num_hidden_layers=2)
    cw = weightgen.make_dit_weights(ccfg.weight_shapes(), ccfg.hidden_size, seed=31, mode="test")
    dw = weightgen.make_dit_weights(dcfg.weight_shapes(), dcfg.hidden_size, seed=32, mode="test")
    null = weightgen.make_null_condition_emb(dcfg.hidden_size, seed=32)
    g = torch.Generator().manual_seed(33)
    B, Lt, Ll, Tref, T = 1, 12, 40, 30, 60
    text, lyric, refer = torch.randn(B, Lt, 64, generator=g), torch.randn(B, Ll, 64, generator=g), torch.randn(1, Tref, 64, generator=g)
    tmask = (torch.arange(Lt)[None, :] < 9).long()
    lmask = (torch.arange(Ll)[None, :] < 25).long()
    order = torch.tensor([0])
    ctx = torch.cat([0.5 * torch.randn(B, T, 64, generator=g), torch.ones(B, T, 64)], -1)
    kw = dict(seed=[77], infer_steps=6, diffusion_guidance_sale=7.0)
    # oracle chain
    o_enc, o_mask = o_cond.condition_encoder(o_cond.CondConfig(**tiny, text_hidden_dim=64, timbre_hidden_dim=64, num_lyric_encoder_hidden_layers=2,
                                                               num_timbre_encoder_hidden_layers=2), cw, text, tmask, lyric, lmask, refer, order)
    o_lat = o_sampler.generate_audio(o_dit.DitConfig(**tiny, num_hidden_layers=2), dw, null, o_enc, ctx, **kw)
    # native chain
    enc = NativeCondEncoder(ccfg, gpu_device)
    enc.load_state_dict(cw)
    n_enc, n_mask = enc(text, tmask, lyric, lmask, refer, order)
    assert torch.equal(n_mask.cpu(), o_mask)
    dit = NativeDit(dcfg, gpu_device)
    dit.load_state_dict(dw)
    n_lat = generate_latents(dit, null, n_enc, ctx, **kw)["target_latents"]
    r = _rel(n_lat.cpu(), o_lat)
    print(f"encoder -> sampler chain: latents rel L2 vs the oracle chain {r:.3e}")
    # twice the drift of the oracle CHAIN when both of its stages store weights / contraction operands in bf16 (measured 1.70e-3)
    e_enc, _m = _drift.emulated(o_cond.condition_encoder, o_cond.CondConfig(**tiny, text_hidden_dim=64, timbre_hidden_dim=64, num_lyric_encoder_hidden_layers=2,
                                                                          num_timbre_encoder_hidden_layers=2), cw, text, tmask, lyric, lmask, refer, order)
    e_lat = _drift.emulated(o_sampler.generate_audio, o_dit.DitConfig(**tiny, num_hidden_layers=2), dw, null, e_enc, ctx, **kw)
    _drift.check("encoder -> sampler chain", r, _rel(e_lat, o_lat))
    vcfg = ace355.VaeConfig()
    vw = weightgen.make_vae_weights(vcfg.weight_shapes(), seed=34, mode="init")
    vae = NativeVae(vcfg, gpu_device)
    vae.load_state_dict(vw)
    wav = vae.decode(n_lat[:, :24].transpose(1, 2).contiguous()).cpu()
    wref = o_vae.decode(o_vae.VaeConfig(), vw, o_lat[:, :24].transpose(1, 2).contiguous())
    snr = float(10 * torch.log10(wref.pow(2).sum() / (wav - wref).pow(2).sum()))
    print(f"encoder -> sampler -> decoder chain: waveform SNR vs the oracle chain {snr:.1f} dB")
    assert snr > 25.0, snr
