"""Architecture constants the native path needs.

Mirrors the fields of the reference's ``AceStepConfig``
(/root/reference/acestep/models/base/configuration_acestep_v15.py:148-263) and of
``AutoencoderOobleck.config`` (acestep/models/mlx/vae_model.py:251-264, 322-336).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple


def _effective_layer_types(cfg) -> List[str]:
    """``layer_types`` as the reference APPLIES them: with ``use_sliding_window`` off (``sliding_window`` None) the
    sliding mask is never built (modeling_acestep_v15_base.py:1397 passes ``sliding_attn_mask=None``), so every layer
    attends fully whatever ``layer_types`` says; the native kernels would read window 0 as a one-key band."""
    lt = list(cfg.layer_types)
    if not getattr(cfg, "use_sliding_window", True) or getattr(cfg, "sliding_window", None) is None:
        return ["full_attention"] * len(lt)
    return lt


@dataclass
class DitConfig:
    hidden_size: int = 2048
    intermediate_size: int = 6144
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    sliding_window: int = 128
    patch_size: int = 2
    in_channels: int = 192
    audio_acoustic_hidden_dim: int = 64
    layer_types: Optional[List[str]] = None

    def __post_init__(self):
        if self.layer_types is None:
            # configuration_acestep_v15.py:251-254
            self.layer_types = [
                "sliding_attention" if (i + 1) % 2 else "full_attention" for i in range(self.num_hidden_layers)
            ]

    @classmethod
    def from_reference(cls, cfg) -> "DitConfig":
        """Build from an ``AceStepConfig`` instance (``self.model.config`` on the handler)."""
        return cls(
            hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
            num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
            num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
            rms_norm_eps=cfg.rms_norm_eps, rope_theta=float(getattr(cfg, "rope_theta", 1e6)),
            sliding_window=cfg.sliding_window or 0, patch_size=cfg.patch_size, in_channels=cfg.in_channels,
            audio_acoustic_hidden_dim=cfg.audio_acoustic_hidden_dim, layer_types=_effective_layer_types(cfg),
        )

    def weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Names/shapes of ``AceStepDiTModel.state_dict()`` (modeling_acestep_v15_base.py:1248-1299)."""
        D, Fh, hd = self.hidden_size, self.intermediate_size, self.head_dim
        q, kv = self.num_attention_heads * hd, self.num_key_value_heads * hd
        s: Dict[str, Tuple[int, ...]] = {}
        for li in range(self.num_hidden_layers):
            p = f"layers.{li}."
            s[p + "scale_shift_table"] = (1, 6, D)
            for n in ("self_attn_norm", "cross_attn_norm", "mlp_norm"):
                s[p + n + ".weight"] = (D,)
            for a in ("self_attn", "cross_attn"):
                s[p + a + ".q_proj.weight"] = (q, D)
                s[p + a + ".k_proj.weight"] = (kv, D)
                s[p + a + ".v_proj.weight"] = (kv, D)
                s[p + a + ".o_proj.weight"] = (D, q)
                s[p + a + ".q_norm.weight"] = (hd,)
                s[p + a + ".k_norm.weight"] = (hd,)
            s[p + "mlp.gate_proj.weight"] = (Fh, D)
            s[p + "mlp.up_proj.weight"] = (Fh, D)
            s[p + "mlp.down_proj.weight"] = (D, Fh)
        s["proj_in.1.weight"] = (D, self.in_channels, self.patch_size)
        s["proj_in.1.bias"] = (D,)
        for e in ("time_embed", "time_embed_r"):
            s[e + ".linear_1.weight"] = (D, 256)
            s[e + ".linear_1.bias"] = (D,)
            s[e + ".linear_2.weight"] = (D, D)
            s[e + ".linear_2.bias"] = (D,)
            s[e + ".time_proj.weight"] = (6 * D, D)
            s[e + ".time_proj.bias"] = (6 * D,)
        s["condition_embedder.weight"] = (D, D)
        s["condition_embedder.bias"] = (D,)
        s["norm_out.weight"] = (D,)
        s["proj_out.1.weight"] = (D, self.audio_acoustic_hidden_dim, self.patch_size)
        s["proj_out.1.bias"] = (self.audio_acoustic_hidden_dim,)
        s["scale_shift_table"] = (1, 2, D)
        return s


@dataclass
class CondConfig:
    """Fields of ``AceStepConfig`` used by ``AceStepConditionEncoder`` (modeling_acestep_v15_base.py:1509-1554)."""

    hidden_size: int = 2048
    intermediate_size: int = 6144
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    sliding_window: int = 128
    text_hidden_dim: int = 1024
    timbre_hidden_dim: int = 64
    num_lyric_encoder_hidden_layers: int = 8
    num_timbre_encoder_hidden_layers: int = 4
    layer_types: Optional[List[str]] = None

    def __post_init__(self):
        if self.layer_types is None:
            n = max(self.num_lyric_encoder_hidden_layers, self.num_timbre_encoder_hidden_layers)
            self.layer_types = ["sliding_attention" if (i + 1) % 2 else "full_attention" for i in range(n)]

    @classmethod
    def from_reference(cls, cfg) -> "CondConfig":
        return cls(
            hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
            num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_norm_eps,
            rope_theta=float(getattr(cfg, "rope_theta", 1e6)), sliding_window=cfg.sliding_window or 0,
            text_hidden_dim=cfg.text_hidden_dim, timbre_hidden_dim=cfg.timbre_hidden_dim,
            num_lyric_encoder_hidden_layers=cfg.num_lyric_encoder_hidden_layers,
            num_timbre_encoder_hidden_layers=cfg.num_timbre_encoder_hidden_layers, layer_types=_effective_layer_types(cfg),
        )

    def weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Names/shapes of ``AceStepConditionEncoder.state_dict()`` (608 M parameters at the default config)."""
        D, Fh, hd = self.hidden_size, self.intermediate_size, self.head_dim
        q, kv = self.num_attention_heads * hd, self.num_key_value_heads * hd
        s: Dict[str, Tuple[int, ...]] = {"text_projector.weight": (D, self.text_hidden_dim)}
        for p, n, din in (("lyric_encoder.", self.num_lyric_encoder_hidden_layers, self.text_hidden_dim),
                          ("timbre_encoder.", self.num_timbre_encoder_hidden_layers, self.timbre_hidden_dim)):
            s[p + "embed_tokens.weight"] = (D, din)
            s[p + "embed_tokens.bias"] = (D,)
            s[p + "norm.weight"] = (D,)
            if p == "timbre_encoder.":
                s[p + "special_token"] = (1, 1, D)
            for li in range(n):
                r = f"{p}layers.{li}."
                s[r + "self_attn.q_proj.weight"] = (q, D)
                s[r + "self_attn.k_proj.weight"] = (kv, D)
                s[r + "self_attn.v_proj.weight"] = (kv, D)
                s[r + "self_attn.o_proj.weight"] = (D, q)
                s[r + "self_attn.q_norm.weight"] = (hd,)
                s[r + "self_attn.k_norm.weight"] = (hd,)
                s[r + "input_layernorm.weight"] = (D,)
                s[r + "post_attention_layernorm.weight"] = (D,)
                s[r + "mlp.gate_proj.weight"] = (Fh, D)
                s[r + "mlp.up_proj.weight"] = (Fh, D)
                s[r + "mlp.down_proj.weight"] = (D, Fh)
        return s


@dataclass
class DetokConfig:
    """Fields of ``AceStepConfig`` used by ``AudioTokenDetokenizer`` (modeling_acestep_v15_base.py:862-994)."""

    hidden_size: int = 2048
    intermediate_size: int = 6144
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    sliding_window: int = 128
    pool_window_size: int = 5
    num_attention_pooler_hidden_layers: int = 2
    audio_acoustic_hidden_dim: int = 64
    layer_types: Optional[List[str]] = None

    def __post_init__(self):
        if self.layer_types is None:
            self.layer_types = ["sliding_attention" if (i + 1) % 2 else "full_attention"
                                for i in range(self.num_attention_pooler_hidden_layers)]

    @classmethod
    def from_reference(cls, cfg) -> "DetokConfig":
        return cls(
            hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
            num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_norm_eps,
            rope_theta=float(getattr(cfg, "rope_theta", 1e6)), sliding_window=cfg.sliding_window or 0,
            pool_window_size=cfg.pool_window_size, num_attention_pooler_hidden_layers=cfg.num_attention_pooler_hidden_layers,
            audio_acoustic_hidden_dim=cfg.audio_acoustic_hidden_dim,
            layer_types=_effective_layer_types(cfg)[: cfg.num_attention_pooler_hidden_layers],
        )

    def weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Names/shapes of ``AudioTokenDetokenizer.state_dict()``."""
        D, Fh, hd = self.hidden_size, self.intermediate_size, self.head_dim
        q, kv = self.num_attention_heads * hd, self.num_key_value_heads * hd
        s: Dict[str, Tuple[int, ...]] = {
            "embed_tokens.weight": (D, D), "embed_tokens.bias": (D,), "norm.weight": (D,),
            "special_tokens": (1, self.pool_window_size, D), "proj_out.weight": (self.audio_acoustic_hidden_dim, D),
            "proj_out.bias": (self.audio_acoustic_hidden_dim,)}
        for li in range(self.num_attention_pooler_hidden_layers):
            r = f"layers.{li}."
            s[r + "self_attn.q_proj.weight"] = (q, D)
            s[r + "self_attn.k_proj.weight"] = (kv, D)
            s[r + "self_attn.v_proj.weight"] = (kv, D)
            s[r + "self_attn.o_proj.weight"] = (D, q)
            s[r + "self_attn.q_norm.weight"] = (hd,)
            s[r + "self_attn.k_norm.weight"] = (hd,)
            s[r + "input_layernorm.weight"] = (D,)
            s[r + "post_attention_layernorm.weight"] = (D,)
            s[r + "mlp.gate_proj.weight"] = (Fh, D)
            s[r + "mlp.up_proj.weight"] = (Fh, D)
            s[r + "mlp.down_proj.weight"] = (D, Fh)
        return s


@dataclass
class VaeConfig:
    """Decoder half of AutoencoderOobleck.  Strides are run-time data (checkpoints/vae/config.json);
    the synthetic default has hop 1920 (handler/conditioning_target.py:47,53)."""

    decoder_channels: int = 128
    decoder_input_channels: int = 64
    audio_channels: int = 2
    channel_multiples: Tuple[int, ...] = (1, 2, 4, 8, 16)
    downsampling_ratios: Tuple[int, ...] = (2, 4, 4, 6, 10)

    @classmethod
    def from_reference(cls, cfg) -> "VaeConfig":
        return cls(decoder_channels=cfg.decoder_channels, decoder_input_channels=cfg.decoder_input_channels,
                   audio_channels=cfg.audio_channels, channel_multiples=tuple(cfg.channel_multiples),
                   downsampling_ratios=tuple(cfg.downsampling_ratios))

    @property
    def upsampling_ratios(self) -> Tuple[int, ...]:
        return tuple(self.downsampling_ratios[::-1])

    @property
    def hop(self) -> int:
        return int(math.prod(self.downsampling_ratios))

    def block_dims(self) -> List[Tuple[int, int, int]]:
        cm = [1] + list(self.channel_multiples)
        s = self.upsampling_ratios
        n = len(s)
        return [(self.decoder_channels * cm[n - i], self.decoder_channels * cm[n - i - 1], s[i]) for i in range(n)]

    def weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """state_dict names/shapes of the decoder half (acestep/models/mlx/vae_convert.py:62-127)."""
        sh: Dict[str, Tuple[int, ...]] = {}

        def conv(name, cout, cin, k, bias=True):
            sh[name + ".weight_g"] = (cout, 1, 1)
            sh[name + ".weight_v"] = (cout, cin, k)
            if bias:
                sh[name + ".bias"] = (cout,)

        def snk(name, c):
            sh[name + ".alpha"] = (1, c, 1)
            sh[name + ".beta"] = (1, c, 1)

        dims = self.block_dims()
        conv("decoder.conv1", dims[0][0], self.decoder_input_channels, 7)
        for i, (cin, cout, s) in enumerate(dims):
            p = f"decoder.block.{i}"
            snk(p + ".snake1", cin)
            sh[p + ".conv_t1.weight_g"] = (cin, 1, 1)
            sh[p + ".conv_t1.weight_v"] = (cin, cout, 2 * s)
            sh[p + ".conv_t1.bias"] = (cout,)
            for j in (1, 2, 3):
                r = f"{p}.res_unit{j}"
                snk(r + ".snake1", cout)
                conv(r + ".conv1", cout, cout, 7)
                snk(r + ".snake2", cout)
                conv(r + ".conv2", cout, cout, 1)
        snk("decoder.snake1", self.decoder_channels)
        conv("decoder.conv2", self.audio_channels, self.decoder_channels, 7, bias=False)
        return sh

    @property
    def encoder_hidden_size(self) -> int:
        return 2 * self.decoder_input_channels  # mean | scale

    def encoder_block_dims(self) -> List[Tuple[int, int, int]]:
        cm = [1] + list(self.channel_multiples)
        e = self.encoder_hidden_size
        return [(e * cm[i], e * cm[i + 1], s) for i, s in enumerate(self.downsampling_ratios)]

    def encoder_weight_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """state_dict names/shapes of the encoder half (acestep/models/mlx/vae_model.py:92-116, 148-187)."""
        sh: Dict[str, Tuple[int, ...]] = {}

        def conv(name, cout, cin, k):
            sh[name + ".weight_g"] = (cout, 1, 1)
            sh[name + ".weight_v"] = (cout, cin, k)
            sh[name + ".bias"] = (cout,)

        def snk(name, c):
            sh[name + ".alpha"] = (1, c, 1)
            sh[name + ".beta"] = (1, c, 1)

        conv("encoder.conv1", self.encoder_hidden_size, self.audio_channels, 7)
        dims = self.encoder_block_dims()
        for i, (cin, cout, s) in enumerate(dims):
            p = f"encoder.block.{i}"
            for j in (1, 2, 3):
                r = f"{p}.res_unit{j}"
                snk(r + ".snake1", cin)
                conv(r + ".conv1", cin, cin, 7)
                snk(r + ".snake2", cin)
                conv(r + ".conv2", cin, cin, 1)
            snk(p + ".snake1", cin)
            conv(p + ".conv1", cout, cin, 2 * s)
        snk("encoder.snake1", dims[-1][1])
        conv("encoder.conv2", self.encoder_hidden_size, dims[-1][1], 3)
        return sh
