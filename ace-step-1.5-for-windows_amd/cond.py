"""NativeCondEncoder: Python owner of an ``ace355_cond`` handle.

Mirror of ``AceStepConditionEncoder`` (modeling_acestep_v15_base.py:1509-1554): same argument names, same outputs
(``encoder_hidden_states`` with the valid tokens first, ``encoder_attention_mask``).  Weights come from the loaded
``model.encoder.state_dict()``.  Attention masks must be prefix masks (ones then zeros, what the tokenizer's right padding
and ``pack_sequences`` produce); anything else raises ``ValueError``.  Installed through ``modswap.swap_in`` (an ``nn.Module``
wrapper that keeps the reference module and runs it whenever this class raises) - the object itself is not an ``nn.Module`` and
cannot be assigned to ``model.encoder`` directly.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple, Union

import torch

from . import native
from .config import CondConfig


def _prefix_lengths(mask: torch.Tensor, what: str) -> torch.Tensor:
    m = mask.detach().to("cpu").to(torch.bool)
    lens = m.sum(dim=1)
    expect = torch.arange(m.shape[1])[None, :] < lens[:, None]
    if not torch.equal(m, expect):
        raise ValueError(f"ace355: {what} must be a prefix mask (valid tokens first); got an interior hole")
    return lens.to(torch.int32)


class NativeCondEncoder:
    def __init__(self, cfg: CondConfig, device: Union[str, torch.device] = "cuda:0", out_dtype: torch.dtype = torch.float32):
        self.cfg = cfg
        self.device = torch.device(device)
        self.out_dtype = out_dtype
        self._lib = native.lib()
        n = max(cfg.num_lyric_encoder_hidden_layers, cfg.num_timbre_encoder_hidden_layers)
        if len(cfg.layer_types) < n or n > 64:
            raise ValueError("ace355: layer_types shorter than the encoder stacks")
        mask = 0
        for i in range(n):
            if cfg.layer_types[i] == "sliding_attention":
                mask |= 1 << i
        c = native.CondConfigC(cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim,
                               cfg.text_hidden_dim, cfg.timbre_hidden_dim, cfg.num_lyric_encoder_hidden_layers,
                               cfg.num_timbre_encoder_hidden_layers, cfg.sliding_window, mask, cfg.rms_norm_eps, cfg.rope_theta)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_cond_create(C.byref(c), C.byref(h)), "cond_create")
        self._h = h

    @classmethod
    def from_reference(cls, encoder_module, device: Union[str, torch.device], out_dtype: torch.dtype = torch.float32) -> "NativeCondEncoder":
        """Build from the loaded reference module (``handler.model.encoder``): its config and its state_dict."""
        self = cls(CondConfig.from_reference(encoder_module.config), device, out_dtype)
        self.load_state_dict(encoder_module.state_dict())
        return self

    def close(self):
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):
                self._lib.ace355_cond_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Keys of ``AceStepConditionEncoder.state_dict()``; rotary ``inv_freq`` buffers are skipped."""
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if "rotary_emb" in name:
                    continue
                t = t.detach()
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                t = t.contiguous()
                dt = native.DTYPE_F32 if t.dtype == torch.float32 else native.DTYPE_BF16
                native.check(self._lib.ace355_cond_load_tensor(self._h, name.encode(), native.ptr(t), dt, t.numel(),
                                                               1 if t.is_cuda else 0), f"cond_load_tensor({name})")
            native.check(self._lib.ace355_cond_finalize(self._h), "cond_finalize")

    def __call__(self, text_hidden_states: torch.Tensor, text_attention_mask: torch.Tensor, lyric_hidden_states: torch.Tensor,
                 lyric_attention_mask: torch.Tensor, refer_audio_acoustic_hidden_states_packed: torch.Tensor,
                 refer_audio_order_mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Same signature and outputs as ``AceStepConditionEncoder.forward`` (fp32 output on this device)."""
        B, Lt, TD = text_hidden_states.shape
        B2, Ll, TD2 = lyric_hidden_states.shape
        Nref, Tref, AD = refer_audio_acoustic_hidden_states_packed.shape
        if B2 != B or TD != self.cfg.text_hidden_dim or TD2 != TD or AD != self.cfg.timbre_hidden_dim:
            raise ValueError("ace355: condition encoder input shapes do not match the configuration")
        tl = _prefix_lengths(text_attention_mask, "text_attention_mask")
        ll = _prefix_lengths(lyric_attention_mask, "lyric_attention_mask")
        order = refer_audio_order_mask.detach().to("cpu").to(torch.int32).contiguous()
        if order.numel() != Nref or int(order.min()) < 0 or int(order.max()) >= B:
            raise ValueError("ace355: refer_audio_order_mask must hold one batch index in [0, B) per reference clip")
        if int(order.max()) + 1 != B:
            # the reference sizes the timbre batch by order_mask.max()+1 (base.py:1036) and would fail to concatenate
            raise ValueError("ace355: every batch item needs at least one reference clip (as in the reference)")
        i32p = C.POINTER(C.c_int32)
        as_p = lambda t: C.cast(t.data_ptr(), i32p)  # noqa: E731
        lout = int(self._lib.ace355_cond_out_len(Ll, Lt, as_p(order), Nref, B))
        if lout < 0:
            raise ValueError("ace355: bad condition encoder arguments")
        f32 = lambda t: t.detach().to(self.device, torch.float32).contiguous()  # noqa: E731
        text, lyric, refer = f32(text_hidden_states), f32(lyric_hidden_states), f32(refer_audio_acoustic_hidden_states_packed)
        out = torch.empty(B, lout, self.cfg.hidden_size, device=self.device, dtype=torch.float32)
        out_len = torch.zeros(B, dtype=torch.int32)
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_cond_encode(self._h, native.ptr(text), as_p(tl), Lt, native.ptr(lyric), as_p(ll), Ll,
                                                      native.ptr(refer), as_p(order), Nref, Tref, B, native.ptr(out), as_p(out_len),
                                                      native.current_stream_ptr()), "cond_encode")
        mask = (torch.arange(lout)[None, :] < out_len[:, None]).to(self.device)
        return (out if self.out_dtype == torch.float32 else out.to(self.out_dtype)), mask

    forward = __call__
