"""LM-hint path: ``<|audio_code_N|>`` text -> code indices -> FSQ output -> detokenizer ->
``lm_hints_25Hz`` (what replaces ``src_latents`` where ``is_covers``; modeling_acestep_v15_base.py:1638-1649) - and the other
direction for cover tasks without precomputed hints (base.py:1645): source latents -> ``NativeAudioTokenizer`` (acoustic projection +
attention pooler natively, FSQ in torch) -> quantised 5 Hz tokens -> detokenizer.

Mirror of ``AudioCodesMixin._parse_audio_code_string`` / ``_decode_audio_codes_to_latents``
(acestep/core/generation/handler/audio_codes.py:20-66).  ``NativeDetokenizer`` is call-compatible with the reference's
``AudioTokenDetokenizer`` module (``model.detokenizer``).  The FSQ's two projections ([n, 6] x [6, 2048] and its transpose) run on the
library's own fp32 kernel (``ace355_linear_f32``; no vendor GEMM on this path since round 6), the digit arithmetic around them is a few
elementwise torch ops on the same device; all of it belongs to ``vector_quantize_pytorch`` (absent here): the restatements in
``fsq_output_from_indices`` / ``fsq_quantize`` are parity-unpinned.
"""
from __future__ import annotations

import ctypes as C
import re
from typing import Dict, List, Optional, Sequence, Union

import torch

from . import native
from .config import DetokConfig

MAX_AUDIO_CODE = 63999
FSQ_LEVELS = (8, 8, 8, 5, 5, 5)


def _linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """``F.linear`` in fp32 on the library's kernel (``ace355_linear_f32``: fp64 accumulation, one summation order): x [..., K],
    weight [N, K] -> [..., N] on weight's device.  There is no CPU path: the weights must live on the GPU."""
    if not weight.is_cuda:
        raise RuntimeError("ace355: the FSQ projections run on the native library; move the quantizer weights to the GPU")
    dev = weight.device
    w = weight.detach().to(torch.float32).contiguous()
    b = None if bias is None else bias.detach().to(dev, torch.float32).contiguous()
    x2 = x.detach().to(dev, torch.float32).reshape(-1, x.shape[-1]).contiguous()
    N, K = w.shape
    if x2.shape[1] != K:
        raise ValueError("ace355: FSQ projection input width does not match the weight")
    out = torch.empty(x2.shape[0], N, device=dev, dtype=torch.float32)
    if x2.shape[0]:
        with torch.cuda.device(dev):
            native.check(native.lib().ace355_linear_f32(native.ptr(x2), native.ptr(w), native.ptr(b), native.ptr(out), x2.shape[0], N, K,
                                                        native.current_stream_ptr()), "linear_f32")
    return out.reshape(*x.shape[:-1], N)


def parse_audio_code_string(code_str: str) -> List[int]:
    """Every ``<|audio_code_N|>`` of the string, clamped to [0, 63999] (audio_codes.py:20-47)."""
    if not code_str:
        return []
    return [max(0, min(int(x), MAX_AUDIO_CODE)) for x in re.findall(r"<\|audio_code_(\d+)\|>", code_str)]


def fsq_output_from_indices(indices: torch.Tensor, project_out_weight: torch.Tensor, project_out_bias: Optional[torch.Tensor] = None,
                            levels: Sequence[int] = FSQ_LEVELS) -> torch.Tensor:
    """ResidualFSQ(num_quantizers=1).get_output_from_indices restated: indices [B, T, 1] (or [B, T]) -> [B, T, dim]."""
    idx = (indices[..., 0] if indices.dim() == 3 else indices).to(project_out_weight.device)
    lv = torch.tensor(list(levels), dtype=torch.int64, device=idx.device)
    basis = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.int64, device=idx.device), lv[:-1]]), dim=0)
    digits = (idx.to(torch.int64).unsqueeze(-1) // basis) % lv
    half = (lv // 2).to(torch.float32)
    codes = (digits.to(torch.float32) - half) / half
    return _linear(codes, project_out_weight, project_out_bias)


class NativeDetokenizer:
    def __init__(self, cfg: DetokConfig, device: Union[str, torch.device] = "cuda:0", out_dtype: torch.dtype = torch.float32):
        self.cfg = cfg
        self.device = torch.device(device)
        self.out_dtype = out_dtype
        self._lib = native.lib()
        mask = 0
        for i in range(cfg.num_attention_pooler_hidden_layers):
            if cfg.layer_types[i] == "sliding_attention":
                mask |= 1 << i
        c = native.DetokConfigC(cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                                cfg.num_attention_pooler_hidden_layers, cfg.pool_window_size, cfg.audio_acoustic_hidden_dim,
                                cfg.sliding_window, mask, cfg.rms_norm_eps, cfg.rope_theta)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_detok_create(C.byref(c), C.byref(h)), "detok_create")
        self._h = h

    @classmethod
    def from_reference(cls, module, device: Union[str, torch.device], out_dtype: torch.dtype = torch.float32) -> "NativeDetokenizer":
        """Build from the loaded reference module (``handler.model.detokenizer``)."""
        self = cls(DetokConfig.from_reference(module.config), device, out_dtype)
        self.load_state_dict(module.state_dict())
        return self

    def close(self):
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):
                self._lib.ace355_detok_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if "rotary_emb" in name:
                    continue
                t = t.detach()
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                t = t.contiguous()
                dt = native.DTYPE_F32 if t.dtype == torch.float32 else native.DTYPE_BF16
                native.check(self._lib.ace355_detok_load_tensor(self._h, name.encode(), native.ptr(t), dt, t.numel(),
                                                                1 if t.is_cuda else 0), f"detok_load_tensor({name})")
            native.check(self._lib.ace355_detok_finalize(self._h), "detok_finalize")

    def __call__(self, x: torch.Tensor, attention_mask=None) -> torch.Tensor:
        """x [B, T5, hidden] -> [B, T5 * pool_window_size, 64] (``AudioTokenDetokenizer.forward``; the reference never
        passes a mask here: audio_codes.py:65, base.py:1647)."""
        if attention_mask is not None:
            raise ValueError("ace355: the detokenizer takes no attention mask (the reference never passes one)")
        B, T5, D = x.shape
        if D != self.cfg.hidden_size:
            raise ValueError("ace355: detokenizer input width does not match the configuration")
        x = x.detach().to(self.device, torch.float32).contiguous()
        out = torch.empty(B, T5 * self.cfg.pool_window_size, self.cfg.audio_acoustic_hidden_dim, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_detok_run(self._h, native.ptr(x), B, T5, native.ptr(out), native.current_stream_ptr()), "detok_run")
        return out if self.out_dtype == torch.float32 else out.to(self.out_dtype)

    forward = __call__


def fsq_quantize(z: torch.Tensor, project_in_weight: torch.Tensor, project_in_bias: Optional[torch.Tensor], project_out_weight: torch.Tensor,
                 project_out_bias: Optional[torch.Tensor] = None, levels: Sequence[int] = FSQ_LEVELS, eps: float = 1e-3):
    """ResidualFSQ(num_quantizers=1).forward restated (parity unpinned, like ``fsq_output_from_indices``): z [..., dim] ->
    (quantized [..., dim], indices [..., 1]).  project_in -> bounded tanh (half_l = (L - 1)(1 + eps) / 2, half-step offset for
    even L) -> round -> codes / floor(L / 2) -> project_out; index = mixed-radix number of the shifted digits."""
    dev = project_in_weight.device
    lv = torch.tensor(list(levels), dtype=torch.float32, device=dev)
    y = _linear(z, project_in_weight, project_in_bias)
    half_l = (lv - 1) * (1 + eps) / 2
    offset = torch.where(lv.to(torch.int64) % 2 == 0, torch.full_like(lv, 0.5), torch.zeros_like(lv))
    q = torch.round(torch.tanh(y + torch.atanh(offset / half_l)) * half_l - offset)
    half_w = torch.floor(lv / 2)
    lvi = lv.to(torch.int64)
    basis = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.int64, device=dev), lvi[:-1]]), dim=0)
    idx = ((q + half_w).to(torch.int64) * basis).sum(-1, keepdim=True)
    quantized = _linear(q / half_w, project_out_weight, project_out_bias)
    return quantized, idx


class NativeAudioTokenizer:
    """Call-compatible with the reference's ``AceStepAudioTokenizer`` (``model.tokenizer``, base.py:1181-1223): ``forward(x)`` with
    ``x [B, T5, pool, 64]`` and ``tokenize(x [B, T5 * pool, 64])`` return ``(quantized [B, T5, hidden], indices [B, T5, 1])``.  The
    acoustic projection and the attention pooler run natively (``ace355_tok_*``); the quantizer (``quantizer.project_in / project_out``
    of the state dict) is the [n, hidden] x [hidden, 6] product + rounding of ``fsq_quantize``."""

    def __init__(self, cfg: DetokConfig, device: Union[str, torch.device] = "cuda:0", out_dtype: torch.dtype = torch.float32):
        self.cfg = cfg
        self.device = torch.device(device)
        self.out_dtype = out_dtype
        self.pool_window_size = cfg.pool_window_size
        self._lib = native.lib()
        mask = 0
        for i in range(cfg.num_attention_pooler_hidden_layers):
            if cfg.layer_types[i] == "sliding_attention":
                mask |= 1 << i
        c = native.DetokConfigC(cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                                cfg.num_attention_pooler_hidden_layers, cfg.pool_window_size, cfg.audio_acoustic_hidden_dim,
                                cfg.sliding_window, mask, cfg.rms_norm_eps, cfg.rope_theta)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_tok_create(C.byref(c), C.byref(h)), "tok_create")
        self._h = h
        self._q: Dict[str, torch.Tensor] = {}

    @classmethod
    def from_reference(cls, module, device: Union[str, torch.device], out_dtype: torch.dtype = torch.float32) -> "NativeAudioTokenizer":
        self = cls(DetokConfig.from_reference(module.config), device, out_dtype)
        self.load_state_dict(module.state_dict())
        return self

    def close(self):
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):
                self._lib.ace355_tok_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Keys of ``AceStepAudioTokenizer.state_dict()``; ``quantizer.*`` entries are kept for ``fsq_quantize`` (only the two
        projections are read: ``quantizer.project_in.{weight,bias}``, ``quantizer.project_out.{weight,bias}``)."""
        self._q = {}
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if "rotary_emb" in name:
                    continue
                t = t.detach()
                if name.startswith("quantizer."):
                    self._q[name[len("quantizer."):]] = t.float().to(self.device)
                    continue
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                t = t.contiguous()
                dt = native.DTYPE_F32 if t.dtype == torch.float32 else native.DTYPE_BF16
                native.check(self._lib.ace355_tok_load_tensor(self._h, name.encode(), native.ptr(t), dt, t.numel(),
                                                              1 if t.is_cuda else 0), f"tok_load_tensor({name})")
            native.check(self._lib.ace355_tok_finalize(self._h), "tok_finalize")

    def pool(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, T5, pool, 64] or [B, T5 * pool, 64] -> pooled [B, T5, hidden] fp32 (everything before the quantizer)."""
        P, A = self.cfg.pool_window_size, self.cfg.audio_acoustic_hidden_dim
        if x.dim() == 4:
            if x.shape[2] != P:
                raise ValueError("ace355: the tokenizer's windows must hold pool_window_size frames")
            x = x.reshape(x.shape[0], x.shape[1] * P, x.shape[3])
        B, T, Ax = x.shape
        if Ax != A or T % P != 0:
            raise ValueError("ace355: tokenizer input must be [B, T5 * pool_window_size, audio_acoustic_hidden_dim]")
        x = x.detach().to(self.device, torch.float32).contiguous()
        out = torch.empty(B, T // P, self.cfg.hidden_size, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_tok_run(self._h, native.ptr(x), B, T // P, native.ptr(out), native.current_stream_ptr()), "tok_run")
        return out

    def __call__(self, hidden_states: torch.Tensor):
        pooled = self.pool(hidden_states)
        q = self._q
        if "project_in.weight" not in q or "project_out.weight" not in q:
            raise RuntimeError("ace355: the tokenizer's quantizer weights (quantizer.project_in / project_out) were not loaded")
        quantized, idx = fsq_quantize(pooled, q["project_in.weight"], q.get("project_in.bias"), q["project_out.weight"], q.get("project_out.bias"))
        return (quantized if self.out_dtype == torch.float32 else quantized.to(self.out_dtype)), idx

    forward = __call__

    def tokenize(self, x: torch.Tensor):
        """``AceStepAudioTokenizer.tokenize`` (base.py:1220-1223): x [B, T5 * pool, 64]."""
        return self(x)


def decode_audio_codes_to_latents(code_str: str, detokenizer, project_out_weight: torch.Tensor,
                                  project_out_bias: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """``_decode_audio_codes_to_latents`` (audio_codes.py:49-66): None when the string holds no codes."""
    ids = parse_audio_code_string(code_str)
    if not ids:
        return None
    dev = project_out_weight.device
    idx = torch.tensor(ids, dtype=torch.long, device=dev).unsqueeze(0).unsqueeze(-1)
    return detokenizer(fsq_output_from_indices(idx, project_out_weight, project_out_bias))
