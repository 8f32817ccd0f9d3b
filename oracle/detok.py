"""fp32 CPU restatement of the LM-hint path (test oracle):
``<|audio_code_N|>`` text -> code indices -> ResidualFSQ.get_output_from_indices -> AudioTokenDetokenizer -> lm_hints_25Hz.

Reference: acestep/core/generation/handler/audio_codes.py:20-66 (parse + call order) and
acestep/models/base/modeling_acestep_v15_base.py:862-994 (``AudioTokenDetokenizer``; ``base.py`` below).

Pinning status:
  * ``parse_audio_code_string`` and ``detokenizer`` are PINNED against the imported reference by
    tests/golden/make_golden.py (fixture G8).
  * ``fsq_output_from_indices`` is **parity unpinned**: the arithmetic lives in the third-party package
    ``vector_quantize_pytorch>=1.27.15`` (pyproject.toml of the reference; class ``ResidualFSQ`` with ``num_quantizers=1``,
    ``levels=[8,8,8,5,5,5]``, ``dim=2048``), which is absent from /root/reference and from this image.  It is restated from
    the published algorithm (Mentzer et al., "Finite Scalar Quantization", and the package's ``FSQ.indices_to_codes`` /
    ``ResidualFSQ.get_output_from_indices``): mixed-radix digits of the index, each digit d_i in [0, L_i) mapped to
    (d_i - floor(L_i/2)) / floor(L_i/2), then the quantizer's ``project_out`` Linear(6 -> 2048).

Test infrastructure only: nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import cond as o_cond
from . import dit as o_dit

Tensor = torch.Tensor

MAX_AUDIO_CODE = 63999  # prod([8,8,8,5,5,5]) - 1 (audio_codes.py:25)


def parse_audio_code_string(code_str: str) -> List[int]:
    """AudioCodesMixin._parse_audio_code_string, audio_codes.py:20-47: every ``<|audio_code_N|>``, clamped to [0, 63999]."""
    if not code_str:
        return []
    return [max(0, min(int(x), MAX_AUDIO_CODE)) for x in re.findall(r"<\|audio_code_(\d+)\|>", code_str)]


def fsq_codes_from_indices(indices: Tensor, levels: Sequence[int]) -> Tensor:
    """FSQ.indices_to_codes without the projection: [...,] int64 -> [..., len(levels)] floats in [-1, 1].

    basis_i = prod(levels[:i]); digit_i = (index // basis_i) % L_i; code_i = (digit_i - L_i // 2) / (L_i // 2).
    """
    lv = torch.tensor(list(levels), dtype=torch.int64)
    basis = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.int64), lv[:-1]]), dim=0)
    digits = (indices.unsqueeze(-1) // basis) % lv
    half = (lv // 2).to(torch.float32)
    return (digits.to(torch.float32) - half) / half


def fsq_output_from_indices(indices: Tensor, levels: Sequence[int], project_out_w: Tensor, project_out_b: Optional[Tensor]) -> Tensor:
    """ResidualFSQ(num_quantizers=1).get_output_from_indices: indices [B, T, 1] -> [B, T, dim] (parity unpinned, see header)."""
    codes = fsq_codes_from_indices(indices[..., 0], levels)
    return F.linear(codes, project_out_w, project_out_b)


@dataclass
class DetokConfig:
    """Fields of AceStepConfig used by AudioTokenDetokenizer (configuration_acestep_v15.py:148-263)."""

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


def detokenizer(cfg: DetokConfig, w: Dict[str, Tensor], x: Tensor) -> Tensor:
    """AudioTokenDetokenizer.forward, base.py:886-994: x [B, T5, D] -> lm_hints [B, T5 * pool, 64].

    embed_tokens (Linear + bias) -> repeat each token `pool` times + special_tokens -> sequences of `pool` tokens through
    the encoder layers (no padding mask; the band never bites at 5 tokens) -> norm -> proj_out -> unfold.
    """
    B, T, D = x.shape
    P = cfg.pool_window_size
    h = o_dit._linear(x, w["embed_tokens.weight"], w["embed_tokens.bias"])
    h = h.unsqueeze(2).repeat(1, 1, P, 1) + w["special_tokens"].expand(B, T, -1, -1)
    h = h.reshape(B * T, P, D)
    cos, sin = o_dit.rope_cos_sin(P, cfg.head_dim, cfg.rope_theta)
    full = o_cond.mask_4d(P, None, None)
    slide = o_cond.mask_4d(P, None, cfg.sliding_window)
    for li in range(cfg.num_attention_pooler_hidden_layers):
        m = slide if cfg.layer_types[li] == "sliding_attention" else full
        h = o_cond.encoder_layer(cfg, w, f"layers.{li}.", h, cos, sin, m)
    h = o_dit.rms_norm(h, w["norm.weight"], cfg.rms_norm_eps)
    h = o_dit._linear(h, w["proj_out.weight"], w["proj_out.bias"])
    return h.reshape(B, T * P, -1)


def detok_weight_shapes(cfg: DetokConfig) -> Dict[str, tuple]:
    """Names/shapes of AudioTokenDetokenizer.state_dict() (``model.detokenizer`` of the reference checkpoint)."""
    D, F_, H, KV, hd = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    s = {"embed_tokens.weight": (D, D), "embed_tokens.bias": (D,), "norm.weight": (D,),
         "special_tokens": (1, cfg.pool_window_size, D), "proj_out.weight": (cfg.audio_acoustic_hidden_dim, D),
         "proj_out.bias": (cfg.audio_acoustic_hidden_dim,)}
    for li in range(cfg.num_attention_pooler_hidden_layers):
        q = f"layers.{li}."
        s[q + "self_attn.q_proj.weight"] = (H * hd, D)
        s[q + "self_attn.k_proj.weight"] = (KV * hd, D)
        s[q + "self_attn.v_proj.weight"] = (KV * hd, D)
        s[q + "self_attn.o_proj.weight"] = (D, H * hd)
        s[q + "self_attn.q_norm.weight"] = (hd,)
        s[q + "self_attn.k_norm.weight"] = (hd,)
        s[q + "input_layernorm.weight"] = (D,)
        s[q + "post_attention_layernorm.weight"] = (D,)
        s[q + "mlp.gate_proj.weight"] = (F_, D)
        s[q + "mlp.up_proj.weight"] = (F_, D)
        s[q + "mlp.down_proj.weight"] = (D, F_)
    return s


# ------------------------------------------------------------------------------------------------ audio tokenizer
# AceStepAudioTokenizer (base.py:1181-1223), the opposite direction: 25 Hz acoustic frames -> 5 Hz quantised tokens.
#   * ``tokenizer_pool`` (audio_acoustic_proj + AttentionPooler, base.py:734-859, 1213-1215) is PINNED against the imported
#     reference by tests/golden/make_golden.py (fixture G17).
#   * ``fsq_quantize`` is **parity unpinned** for the reason given in the header (vector_quantize_pytorch absent): restated from
#     the published algorithm - ResidualFSQ(num_quantizers=1): project_in Linear(dim -> 6); FSQ.bound: half_l = (L - 1)(1 + eps)/2
#     with eps = 1e-3, offset = 0.5 for even L, shift = atanh(offset / half_l), z_b = tanh(z + shift) half_l - offset; round;
#     codes = round(z_b) / floor(L / 2); index = sum((round(z_b) + floor(L/2)) * basis); project_out Linear(6 -> dim).

def tokenizer_pool(cfg: DetokConfig, w: Dict[str, Tensor], x: Tensor) -> Tensor:
    """x [B, T5, P, 64] -> pooled [B, T5, D]: AceStepAudioTokenizer.forward up to the quantizer."""
    B, T, P, _ = x.shape
    D = cfg.hidden_size
    h = o_dit._linear(x, w["audio_acoustic_proj.weight"], w["audio_acoustic_proj.bias"])
    a = "attention_pooler."
    h = o_dit._linear(h, w[a + "embed_tokens.weight"], w[a + "embed_tokens.bias"])
    h = torch.cat([w[a + "special_token"].expand(B, T, 1, -1), h], dim=2).reshape(B * T, P + 1, D)
    S = P + 1
    cos, sin = o_dit.rope_cos_sin(S, cfg.head_dim, cfg.rope_theta)
    full = o_cond.mask_4d(S, None, None)
    slide = o_cond.mask_4d(S, None, cfg.sliding_window)
    for li in range(cfg.num_attention_pooler_hidden_layers):
        m = slide if cfg.layer_types[li] == "sliding_attention" else full
        h = o_cond.encoder_layer(cfg, w, f"{a}layers.{li}.", h, cos, sin, m)
    h = o_dit.rms_norm(h, w[a + "norm.weight"], cfg.rms_norm_eps)
    return h[:, 0, :].reshape(B, T, D)


def fsq_quantize(z: Tensor, levels: Sequence[int], project_in_w: Tensor, project_in_b: Optional[Tensor], project_out_w: Tensor,
                 project_out_b: Optional[Tensor], eps: float = 1e-3):
    """ResidualFSQ(num_quantizers=1).forward: z [..., dim] -> (quantized [..., dim], indices [..., 1])  (parity unpinned)."""
    lv = torch.tensor(list(levels), dtype=torch.float32)
    y = F.linear(z, project_in_w, project_in_b)
    half_l = (lv - 1) * (1 + eps) / 2
    offset = torch.where(lv.to(torch.int64) % 2 == 0, torch.tensor(0.5), torch.tensor(0.0))
    shift = torch.atanh(offset / half_l)
    q = torch.round(torch.tanh(y + shift) * half_l - offset)
    half_w = torch.floor(lv / 2)
    codes = q / half_w
    lvi = lv.to(torch.int64)
    basis = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.int64), lvi[:-1]]), dim=0)
    idx = ((q + half_w).to(torch.int64) * basis).sum(-1, keepdim=True)
    return F.linear(codes, project_out_w, project_out_b), idx


def tok_weight_shapes(cfg: DetokConfig) -> Dict[str, tuple]:
    """Names/shapes of AceStepAudioTokenizer.state_dict() (``model.tokenizer``) outside ``quantizer.``."""
    D = cfg.hidden_size
    s = {"audio_acoustic_proj.weight": (D, cfg.audio_acoustic_hidden_dim), "audio_acoustic_proj.bias": (D,),
         "attention_pooler.embed_tokens.weight": (D, D), "attention_pooler.embed_tokens.bias": (D,),
         "attention_pooler.norm.weight": (D,), "attention_pooler.special_token": (1, 1, D)}
    for k, v in detok_weight_shapes(cfg).items():
        if k.startswith("layers."):
            s["attention_pooler." + k] = v
    return s
