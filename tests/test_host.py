"""CPU: host-side logic of the product package (no GPU, no native compute), in the reference's mixin-host stub style
(handler/diffusion_test.py, handler/vae_decode_mixin_test.py)."""
import math

import pytest
import torch

import ace355
from ace355 import weightgen
from ace355.backend import NativeDitMixin, NativeHandler, NativeVaeMixin
from ace355.dit import prepare_noise, schedule


def test_config_matches_reference_constants():
    c = ace355.DitConfig()
    assert (c.hidden_size, c.intermediate_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, c.head_dim) == \
        (2048, 6144, 24, 16, 8, 128)
    assert c.layer_types[0] == "sliding_attention" and c.layer_types[1] == "full_attention"  # cfg.py:251-254
    shapes = c.weight_shapes()
    assert len(shapes) == 476 and sum(math.prod(s) for s in shapes.values()) == 1575458880  # 1.575 B
    v = ace355.VaeConfig()
    assert v.hop == 1920 and v.upsampling_ratios == (10, 6, 4, 4, 2)
    assert [d[:2] for d in v.block_dims()] == [(2048, 1024), (1024, 512), (512, 256), (256, 128), (128, 128)]


def test_oracle_and_product_agree_on_names():
    from oracle import dit as o_dit
    from oracle import oobleck as o_vae
    assert ace355.DitConfig().weight_shapes() == o_dit.dit_weight_shapes(o_dit.DitConfig())
    assert ace355.VaeConfig().weight_shapes() == o_vae.decoder_weight_shapes(o_vae.VaeConfig())
    assert ace355.VaeConfig().encoder_weight_shapes() == o_vae.encoder_weight_shapes(o_vae.VaeConfig())
    from oracle import cond as o_cond, detok as o_detok
    assert ace355.CondConfig().weight_shapes() == o_cond.cond_weight_shapes(o_cond.CondConfig())
    assert ace355.DetokConfig().weight_shapes() == o_detok.detok_weight_shapes(o_detok.DetokConfig())


def test_weightgen_is_deterministic_and_per_tensor():
    c = ace355.DitConfig(hidden_size=256, intermediate_size=768, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1)
    a = weightgen.make_dit_weights(c.weight_shapes(), 256, seed=3, mode="test")
    one = weightgen.make_dit_weights({"layers.1.mlp.up_proj.weight": (768, 256)}, 256, seed=3, mode="test")
    assert torch.equal(a["layers.1.mlp.up_proj.weight"], one["layers.1.mlp.up_proj.weight"])
    b = weightgen.make_dit_weights(c.weight_shapes(), 256, seed=4, mode="test")
    assert not torch.equal(a["layers.0.mlp.up_proj.weight"], b["layers.0.mlp.up_proj.weight"])
    init = weightgen.make_dit_weights(c.weight_shapes(), 256, seed=3, mode="init")
    assert float(init["proj_in.1.bias"].abs().sum()) == 0 and float(init["norm_out.weight"].sum()) == 256


def test_prepare_noise_follows_reference_cpu_semantics():
    from oracle import sampler as o_sampler
    a = prepare_noise((3, 11, 64), [5, -1, 7])
    b = o_sampler.prepare_noise((3, 11, 64), [5, -1, 7])
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and not torch.equal(a[1], b[1])  # seed < 0 -> random
    assert torch.equal(prepare_noise((2, 4, 64), 9), o_sampler.prepare_noise((2, 4, 64), 9))
    assert schedule(27, 1.0).dtype == torch.float32 and float(schedule(27, 3.0)[0]) == 1.0
    assert torch.equal(schedule(0, 1.0, [1.0, 0.5, 0.0]), torch.tensor([1.0, 0.5, 0.0]))


class _DitHost(NativeDitMixin):
    def __init__(self):
        self.device, self.dtype = "cpu", torch.float32
        self.model = type("M", (), {"null_condition_emb": torch.zeros(1, 1, 8)})()
        self.native_dit = object()


def _args(B=2, T=6, L=3, D=8):
    return dict(encoder_hidden_states=torch.zeros(B, L, D), encoder_attention_mask=None, context_latents=torch.zeros(B, T, 128),
                src_latents=torch.zeros(B, T, 64), seed=[1, 2])


def test_native_run_diffusion_validation_mirrors_mlx_seam():
    h = _DitHost()
    with pytest.raises(ValueError, match="Unsupported infer_method"):
        h._native_run_diffusion(**_args(), infer_method="euler")
    with pytest.raises(TypeError, match="timesteps"):
        h._native_run_diffusion(**_args(), timesteps=3.0)
    bad = _args()
    bad["context_latents"] = torch.zeros(3, 6, 128)
    with pytest.raises(ValueError, match="Batch dimension mismatch"):
        h._native_run_diffusion(**bad)
    bad = _args()
    bad["src_latents"] = torch.zeros(1, 6, 64)
    with pytest.raises(ValueError, match="src_latents"):
        h._native_run_diffusion(**bad)
    with pytest.raises(ValueError, match="non_cover"):
        h._native_run_diffusion(**_args(), encoder_hidden_states_non_cover=torch.zeros(5, 3, 8))
    h2 = _DitHost()
    del h2.native_dit
    NativeDitMixin.native_dit  # class default exists; instance without init is reported as uninitialised
    h2.native_dit = None
    with pytest.raises(RuntimeError, match="not initialised"):
        h2._native_run_diffusion(**_args())
    h3 = _DitHost()
    del h3.dtype
    with pytest.raises(AttributeError, match="dtype"):
        h3._native_run_diffusion(**_args())


def test_native_run_diffusion_routes_turbo_checkpoints(monkeypatch):
    """config.is_turbo (handler init_service_catalog.py:69-73) selects the turbo model's loop: shift -> table, no CFG knobs."""
    import types
    from ace355 import dit as a_dit
    calls = {}

    def fake_turbo(native, enc, ctx, **kw):
        calls["turbo"] = kw
        return {"target_latents": torch.zeros(enc.shape[0], ctx.shape[1], 64), "time_costs": {}}

    def fake_base(native, null, enc, ctx, **kw):
        calls["base"] = kw
        return {"target_latents": torch.zeros(enc.shape[0], ctx.shape[1], 64), "time_costs": {}}

    monkeypatch.setattr(a_dit, "generate_latents_turbo", fake_turbo)
    monkeypatch.setattr(a_dit, "generate_latents", fake_base)
    h = _DitHost()
    h.config = types.SimpleNamespace(is_turbo=True)
    out = h._native_run_diffusion(**_args(), shift=2.0, infer_steps=50, guidance_scale=9.0)
    assert "turbo" in calls and "base" not in calls and calls["turbo"]["shift"] == 2.0 and "infer_steps" not in calls["turbo"]
    assert out["target_latents"].shape == (2, 6, 64)
    h.config = types.SimpleNamespace(is_turbo=False)
    h._native_run_diffusion(**_args(), infer_steps=27)
    assert calls["base"]["infer_steps"] == 27


def test_explicit_timesteps_follow_the_checkpoint_family(monkeypatch):
    """The BASE model's generate_audio has no ``timesteps`` parameter (it lands in **kwargs; the schedule is always linspace +
    shift, base/modeling_acestep_v15_base.py:1812, 1864-1867); sft and turbo read it (sft :1864-1875).  The native seam follows
    the loaded checkpoint: signature probe on ``self.model.generate_audio``, or ``model_variant`` on the host."""
    from ace355 import dit as a_dit
    calls = {}

    def fake_base(native, null, enc, ctx, **kw):
        calls["ts"] = kw["timesteps"]
        return {"target_latents": torch.zeros(enc.shape[0], ctx.shape[1], 64), "time_costs": {}}

    monkeypatch.setattr(a_dit, "generate_latents", fake_base)
    ts = [1.0, 0.6, 0.2, 0.0]

    class BaseModel:
        null_condition_emb = torch.zeros(1, 1, 8)

        def generate_audio(self, src_latents, seed=None, shift=1.0, **kwargs):  # base-family signature
            pass

    class SftModel(BaseModel):
        def generate_audio(self, src_latents, seed=None, shift=1.0, timesteps=None, **kwargs):  # sft-family signature
            pass

    h = _DitHost()
    h.model = BaseModel()
    h._native_run_diffusion(**_args(), timesteps=ts)
    assert calls["ts"] is None
    h.model = SftModel()
    h._native_run_diffusion(**_args(), timesteps=torch.tensor(ts))
    assert calls["ts"] == pytest.approx(ts)
    h.model = BaseModel()
    h.model_variant = "sft"  # explicit override beats the probe
    h._native_run_diffusion(**_args(), timesteps=ts)
    assert calls["ts"] == ts
    nh = NativeHandler()
    assert nh.model_variant == "sft" and nh._model_honours_timesteps()
    nh.model_variant = "base"
    assert not nh._model_honours_timesteps() and not nh._is_turbo()
    nh.model_variant = "turbo"
    assert nh._is_turbo()
    msg, ok = NativeHandler().initialize_service(ace355.DitConfig(), {}, torch.zeros(1, 1, 8), device="cpu", model_variant="bogus")
    assert not ok and "model_variant" in msg


def test_from_reference_without_sliding_window_means_full_attention():
    """use_sliding_window False / sliding_window None: the reference never builds the band mask (base.py:1397, 1431-1440 ->
    sliding_attn_mask None = full attention) although layer_types still alternates; a window of 0 would be a one-key band."""
    import types
    base = dict(hidden_size=256, intermediate_size=768, num_hidden_layers=4, num_attention_heads=2, num_key_value_heads=1, head_dim=128,
                rms_norm_eps=1e-6, rope_theta=1e6, patch_size=2, in_channels=192, audio_acoustic_hidden_dim=64,
                layer_types=["sliding_attention", "full_attention"] * 2)
    on = ace355.DitConfig.from_reference(types.SimpleNamespace(**base, sliding_window=128, use_sliding_window=True))
    assert on.layer_types == base["layer_types"] and on.sliding_window == 128
    for off in (dict(sliding_window=None, use_sliding_window=False), dict(sliding_window=None), dict(sliding_window=128, use_sliding_window=False)):
        c = ace355.DitConfig.from_reference(types.SimpleNamespace(**base, **off))
        assert c.layer_types == ["full_attention"] * 4, off


def test_module_swap_on_a_real_nn_module_parent():
    """model.encoder / model.detokenizer are registered child modules (base.py:1571-1573): a plain object cannot be assigned
    there (TypeError) and nothing around prepare_condition catches a native failure.  NativeModuleSwap is assignable, answers
    state_dict() with the reference's keys and falls back to the kept module when the native call raises."""
    from ace355.modswap import NativeModuleSwap, swap_in

    class Enc(torch.nn.Linear):
        def forward(self, x, mask=None):
            return super().forward(x)

    class Parent(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = Enc(4, 3)

    class FakeNative:
        calls = 0

        def __init__(self, ref):
            self.w = ref.weight.detach().clone()

        @classmethod
        def from_reference(cls, ref, device, out_dtype):
            return cls(ref)

        def __call__(self, x, mask=None):
            if mask is not None:
                raise ValueError("prefix masks only")
            FakeNative.calls += 1
            return x @ self.w.t() * 2.0  # distinguishable from the reference path

    p = Parent()
    keys = set(p.state_dict())
    with pytest.raises(TypeError):
        p.encoder = FakeNative(p.encoder)  # what INTEGRATION.md used to suggest
    ref_mod = p.encoder
    w = swap_in(p, "encoder", FakeNative, "cpu")
    assert isinstance(p.encoder, NativeModuleSwap) and p.encoder is w
    assert set(p.state_dict()) == keys
    x = torch.randn(2, 4)
    assert torch.allclose(p.encoder(x), x @ ref_mod.weight.t() * 2.0) and FakeNative.calls == 1 and w.native_failures == 0
    assert torch.allclose(p.encoder(x, mask=torch.ones(2)), ref_mod(x)) and w.native_failures == 1  # fell back, did not raise
    assert p.encoder.in_features == 4  # attribute passthrough to the module it stands in for
    w2 = swap_in(p, "encoder", FakeNative, "cpu")  # re-entry keeps the ORIGINAL module as the fallback
    assert w2._reference is ref_mod
    # the kept module is a real sub-module: strict load through the PARENT, dtype / device moves and parameters() reach it
    sd = {k: v.clone() + 1.0 for k, v in p.state_dict().items()}
    res = p.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(ref_mod.weight, sd["encoder.weight"]) and set(p.state_dict()) == keys
    p.to(torch.float64)
    assert ref_mod.weight.dtype == torch.float64 and sum(1 for _ in p.parameters()) == 2
    p.to(torch.float32)
    assert torch.allclose(p.encoder(x, mask=torch.ones(2)), ref_mod(x))


def test_init_native_refuses_lora_quant_offload_and_never_raises():
    for flag in ("use_lora", "quantization", "offload_to_cpu"):
        h = _DitHost()
        setattr(h, flag, True if flag != "quantization" else "int8_weight_only")
        assert h._init_native_dit() is False and h.use_native_dit is False and h.native_dit is None
    h = _DitHost()
    h.model = object()  # no config / decoder: failure is swallowed like handler/mlx_dit_init.py:36-43
    assert h._init_native_dit() is False


class _VaeHost(NativeVaeMixin):
    def __init__(self):
        self.device = "cpu"
        self.calls = []

    def _get_auto_decode_chunk_size(self):
        return 512

    def _should_offload_wav_to_cpu(self):
        return False

    def _tiled_decode_inner(self, latents, chunk_size, overlap, offload):
        self.calls.append((chunk_size, overlap, offload))
        return torch.ones(latents.shape[0], 2, latents.shape[-1] * 2)


def test_tiled_decode_native_first_then_pytorch_fallback():
    h = _VaeHost()
    out = h.tiled_decode(torch.zeros(1, 64, 5))  # native not initialised -> host PyTorch path (vae_decode.py:50-85)
    assert out.shape == (1, 2, 10) and h.calls == [(512, 64, False)]
    h.use_native_vae, h.native_vae = True, type("V", (), {"decode": lambda self, z: (_ for _ in ()).throw(RuntimeError("boom"))})()
    out = h.tiled_decode(torch.zeros(1, 64, 5), chunk_size=128, overlap=16)  # failure -> fallback (vae_decode.py:44-48)
    assert out.shape == (1, 2, 10) and h.calls[-1] == (128, 16, False)
    h.native_vae = type("V", (), {"decode": lambda self, z: torch.full((1, 2, 7), 3.0)})()
    assert float(h.tiled_decode(torch.zeros(1, 64, 5)).mean()) == 3.0 and len(h.calls) == 2


def test_tiled_encode_native_first_then_pytorch_fallback():
    """handler/vae_encode.py:28-45 seam: native encode when the encoder half is loaded, else / on failure the host's own."""
    from ace355.backend import NativeVaeMixin

    class _Base:
        def tiled_encode(self, audio, chunk_size=None, overlap=None, offload_latent_to_cpu=True):
            self.calls.append((chunk_size, overlap, offload_latent_to_cpu))
            return torch.zeros(audio.shape[0], 64, 3) if audio.dim() == 3 else torch.zeros(64, 3)

    class _Host(NativeVaeMixin, _Base):
        def __init__(self):
            self.calls = []

    h = _Host()
    assert h.tiled_encode(torch.zeros(1, 2, 100)).shape == (1, 64, 3) and h.calls == [(None, None, True)]
    ok = type("V", (), {"has_encoder": True, "encode": lambda self, a: torch.full((a.shape[0], 64, 5), 2.0)})()
    h.use_native_vae, h.native_vae = True, ok
    assert float(h.tiled_encode(torch.zeros(2, 2, 100)).mean()) == 2.0 and len(h.calls) == 1
    assert h.tiled_encode(torch.zeros(2, 100)).shape == (64, 5)          # 2-D input convention of the reference
    h.native_vae = type("V", (), {"has_encoder": True, "encode": lambda self, a: (_ for _ in ()).throw(RuntimeError("boom"))})()
    assert h.tiled_encode(torch.zeros(1, 2, 100), chunk_size=7).shape == (1, 64, 3) and h.calls[-1] == (7, None, True)
    h.native_vae = type("V", (), {"has_encoder": False})()
    h.tiled_encode(torch.zeros(1, 2, 100))
    assert len(h.calls) == 3


def test_generate_music_never_raises_and_returns_reference_payload_shape():
    h = NativeHandler()  # not initialised: no native backend
    res = h.generate_music(torch.zeros(1, 3, 8), torch.zeros(1, 4, 128), seed=[1])
    assert res["success"] is False and res["audios"] == [] and res["extra_outputs"] == {} and isinstance(res["error"], str)
    assert set(res) == {"audios", "status_message", "extra_outputs", "success", "error"}  # handler/generate_music.py:181-190
    msg, ok = h.initialize_service(ace355.DitConfig(), {}, torch.zeros(1, 1, 2048), device="cpu")
    assert ok is False and "GPU" in msg
    with pytest.raises(ValueError, match="per-call cap"):
        h.service_generate(torch.zeros(9, 3, 8), torch.zeros(9, 4, 128))


def test_flop_model_matches_survey():
    import bench
    S, L = 375, 769
    f = bench.dit_flops_per_forward_per_seq(ace355.DitConfig(), S, L)
    assert abs(f / 1e12 - 1.136) < 0.002  # SURVEY 8d / BASELINE.md section 4
    assert abs(bench.vae_flops_per_frame(ace355.VaeConfig()) / 1e9 - 4.874) < 0.002
    assert abs(bench.dit_flops_per_forward_per_seq(ace355.DitConfig(), 125, L) / 1e12 - 0.375) < 0.001


def test_torch_library_ops_are_registered_with_fake_impls():
    """north_star: "via PyTorch-ROCm custom ops".  torch.ops.ace355.* exist, propagate shapes under FakeTensorMode without a GPU
    or a library call, and have NO CPU kernel (a CPU tensor is refused, never silently computed elsewhere)."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from ace355 import ops
    for name in ("dit_forward", "dit_sample", "vae_decode", "vae_encode", "peak_normalize"):
        assert hasattr(torch.ops.ace355, name), name
    with FakeTensorMode():
        z = torch.empty(3, 64, 100, device="cuda")
        w = torch.ops.ace355.vae_decode(5, z, 1920)
        assert tuple(w.shape) == (3, 2, 192000) and w.dtype == torch.float32 and w.device.type == "cuda"
        a = torch.empty(2, 2, 1920 * 7, device="cuda")
        assert tuple(torch.ops.ace355.vae_encode(5, a, 1920).shape) == (2, 64, 7)
        xt, ctx = torch.empty(2, 50, 64, device="cuda"), torch.empty(2, 50, 128, device="cuda")
        assert tuple(torch.ops.ace355.dit_sample(7, xt, ctx, torch.empty(28)).shape) == (2, 50, 64)
        assert tuple(torch.ops.ace355.dit_forward(7, xt, ctx, [0.5, 0.5], [0.5, 0.5], [0, 1]).shape) == (2, 50, 64)
    with pytest.raises(NotImplementedError):
        torch.ops.ace355.peak_normalize(torch.zeros(1, 2, 4))
    k = ops.register_handle(object())
    assert ops.register_handle(ops._HANDLES[k]) == k
    ops.release_handle(k)
    with pytest.raises(RuntimeError, match="unknown native handle"):
        ops._get(k)
