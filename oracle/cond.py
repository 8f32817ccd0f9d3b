"""fp32 CPU restatement of the ACE-Step 1.5 condition encoder (test oracle).

Reference: /root/reference/acestep/models/base/modeling_acestep_v15_base.py (``base.py`` below):
  AceStepEncoderLayer :374-440, AceStepLyricEncoder :577-731, AceStepTimbreEncoder :997-1178,
  pack_sequences :138-169, AceStepConditionEncoder :1509-1554, create_4d_mask :56-135.
Weight names are the keys of ``AceStepConditionEncoder.state_dict()`` (= ``model.encoder`` of the reference checkpoint).
Pinned by tests/golden/make_golden.py against the imported reference (fixture G7).

Test infrastructure only: nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import dit as o_dit

Tensor = torch.Tensor


@dataclass
class CondConfig:
    """Constants of AceStepConfig (configuration_acestep_v15.py:148-263) the condition encoder uses."""

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
            # configuration_acestep_v15.py:251-254: even index = sliding, odd = full (indexed by the encoder's own layer_idx)
            n = max(self.num_lyric_encoder_hidden_layers, self.num_timbre_encoder_hidden_layers)
            self.layer_types = ["sliding_attention" if (i + 1) % 2 else "full_attention" for i in range(n)]


def mask_4d(seq_len: int, attention_mask: Optional[Tensor], window: Optional[int]) -> Tensor:
    """create_4d_mask(is_causal=False), base.py:56-135: additive [B or 1, 1, S, S], finfo.min where masked.

    valid = band(|i-j| <= window, if sliding) AND key j not padding.  Note the mask is finfo.min, not -inf: a query row with
    NO valid key (a padding query far outside the valid prefix on a sliding layer) softmaxes to uniform over ALL keys.
    """
    valid = o_dit.band_valid(seq_len, window)[None, None]
    if attention_mask is not None:
        valid = valid & attention_mask.view(attention_mask.shape[0], 1, 1, seq_len).to(torch.bool)
    return o_dit.additive_mask(valid)


def encoder_layer(cfg: CondConfig, w: Dict[str, Tensor], p: str, x: Tensor, cos: Tensor, sin: Tensor, mask: Tensor) -> Tensor:
    """AceStepEncoderLayer.forward, base.py:401-440 (pre-norm self-attention + SwiGLU MLP, plain residuals)."""
    eps = cfg.rms_norm_eps
    h = o_dit.rms_norm(x, w[p + "input_layernorm.weight"], eps)
    sp = p + "self_attn."
    q = o_dit.rms_norm(o_dit._heads(o_dit._linear(h, w[sp + "q_proj.weight"]), cfg.head_dim), w[sp + "q_norm.weight"], eps).transpose(1, 2)
    k = o_dit.rms_norm(o_dit._heads(o_dit._linear(h, w[sp + "k_proj.weight"]), cfg.head_dim), w[sp + "k_norm.weight"], eps).transpose(1, 2)
    v = o_dit._heads(o_dit._linear(h, w[sp + "v_proj.weight"]), cfg.head_dim).transpose(1, 2)
    q, k = o_dit.apply_rope(q, k, cos, sin)
    a = o_dit.attention(q, k, v, mask, cfg.head_dim ** -0.5)
    x = x + o_dit._linear(a, w[sp + "o_proj.weight"])
    h = o_dit.rms_norm(x, w[p + "post_attention_layernorm.weight"], eps)
    mp = p + "mlp."
    return x + o_dit._linear(F.silu(o_dit._linear(h, w[mp + "gate_proj.weight"])) * o_dit._linear(h, w[mp + "up_proj.weight"]), w[mp + "down_proj.weight"])


def _encoder_stack(cfg: CondConfig, w: Dict[str, Tensor], p: str, n_layers: int, x: Tensor, attention_mask: Optional[Tensor]) -> Tensor:
    S = x.shape[1]
    cos, sin = o_dit.rope_cos_sin(S, cfg.head_dim, cfg.rope_theta)
    full = mask_4d(S, attention_mask, None)
    slide = mask_4d(S, attention_mask, cfg.sliding_window)
    for li in range(n_layers):
        m = slide if cfg.layer_types[li] == "sliding_attention" else full
        x = encoder_layer(cfg, w, f"{p}layers.{li}.", x, cos, sin, m)
    return o_dit.rms_norm(x, w[p + "norm.weight"], cfg.rms_norm_eps)


def lyric_encoder(cfg: CondConfig, w: Dict[str, Tensor], inputs_embeds: Tensor, attention_mask: Tensor) -> Tensor:
    """AceStepLyricEncoder.forward, base.py:603-731: embed_tokens (Linear with bias) -> 8 layers -> norm."""
    p = "lyric_encoder."
    x = o_dit._linear(inputs_embeds, w[p + "embed_tokens.weight"], w[p + "embed_tokens.bias"])
    return _encoder_stack(cfg, w, p, cfg.num_lyric_encoder_hidden_layers, x, attention_mask)


def unpack_timbre_embeddings(packed: Tensor, order_mask: Tensor) -> Tuple[Tensor, Tensor]:
    """AceStepTimbreEncoder.unpack_timbre_embeddings, base.py:1019-1061: row i of `packed` goes to batch item
    order_mask[i], in order of appearance; returns [B, max_count, d] (zero padded) and its 0/1 mask."""
    N, d = packed.shape
    B = int(order_mask.max().item() + 1)
    counts = torch.bincount(order_mask, minlength=B)
    max_count = int(counts.max().item())
    out = torch.zeros(B, max_count, d, dtype=packed.dtype)
    mask = torch.zeros(B, max_count, dtype=torch.long)
    fill = [0] * B
    for i in range(N):
        b = int(order_mask[i])
        out[b, fill[b]] = packed[i]
        mask[b, fill[b]] = 1
        fill[b] += 1
    return out, mask


def timbre_encoder(cfg: CondConfig, w: Dict[str, Tensor], refer_packed: Tensor, order_mask: Tensor) -> Tuple[Tensor, Tensor]:
    """AceStepTimbreEncoder.forward, base.py:1063-1178: embed -> 4 layers (no padding mask) -> norm -> token 0 -> unpack."""
    p = "timbre_encoder."
    x = o_dit._linear(refer_packed, w[p + "embed_tokens.weight"], w[p + "embed_tokens.bias"])
    x = _encoder_stack(cfg, w, p, cfg.num_timbre_encoder_hidden_layers, x, None)
    return unpack_timbre_embeddings(x[:, 0, :], order_mask)


def pack_sequences(h1: Tensor, h2: Tensor, m1: Tensor, m2: Tensor) -> Tuple[Tensor, Tensor]:
    """pack_sequences, base.py:138-169: concatenate, stable-sort valid tokens first, new mask = arange < count."""
    hc = torch.cat([h1, h2], dim=1)
    mc = torch.cat([m1, m2], dim=1)
    B, L, D = hc.shape
    idx = mc.argsort(dim=1, descending=True, stable=True)
    out = torch.gather(hc, 1, idx.unsqueeze(-1).expand(B, L, D))
    lengths = mc.sum(dim=1)
    new_mask = torch.arange(L)[None, :] < lengths[:, None]
    return out, new_mask


def condition_encoder(cfg: CondConfig, w: Dict[str, Tensor], text_hidden_states: Tensor, text_attention_mask: Tensor,
                      lyric_hidden_states: Tensor, lyric_attention_mask: Tensor, refer_packed: Tensor,
                      refer_order_mask: Tensor) -> Tuple[Tensor, Tensor]:
    """AceStepConditionEncoder.forward, base.py:1526-1554 -> (encoder_hidden_states [B, Ll+Nt+Lt, D], mask)."""
    text = o_dit._linear(text_hidden_states, w["text_projector.weight"])
    lyric = lyric_encoder(cfg, w, lyric_hidden_states, lyric_attention_mask)
    timbre, timbre_mask = timbre_encoder(cfg, w, refer_packed, refer_order_mask)
    enc, mask = pack_sequences(lyric, timbre, lyric_attention_mask, timbre_mask)
    return pack_sequences(enc, text, mask, text_attention_mask)


def cond_weight_shapes(cfg: CondConfig) -> Dict[str, Tuple[int, ...]]:
    """Names/shapes of AceStepConditionEncoder.state_dict() (608 M parameters at the default config)."""
    D, F_, H, KV, hd = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    s: Dict[str, Tuple[int, ...]] = {"text_projector.weight": (D, cfg.text_hidden_dim)}
    for p, n, din in (("lyric_encoder.", cfg.num_lyric_encoder_hidden_layers, cfg.text_hidden_dim),
                      ("timbre_encoder.", cfg.num_timbre_encoder_hidden_layers, cfg.timbre_hidden_dim)):
        s[p + "embed_tokens.weight"] = (D, din)
        s[p + "embed_tokens.bias"] = (D,)
        s[p + "norm.weight"] = (D,)
        if p == "timbre_encoder.":
            s[p + "special_token"] = (1, 1, D)  # parameter of the reference module; unused by its forward
        for li in range(n):
            q = f"{p}layers.{li}."
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
