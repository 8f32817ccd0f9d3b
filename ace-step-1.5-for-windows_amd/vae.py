"""NativeVae: Python owner of an ``ace355_vae`` handle (Oobleck decoder, latent -> waveform; encoder, waveform -> latent).

Mirror of the reference's MLX VAE seam (handler/mlx_vae_init.py:12-96, mlx_vae_decode_native.py:31-72,
models/mlx/vae_convert.py): weights come from the loaded ``AutoencoderOobleck.state_dict()``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Union

import torch

from . import native
from .config import VaeConfig


class NativeVae:
    def __init__(self, cfg: VaeConfig, device: Union[str, torch.device] = "cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        self._lib = native.lib()
        ups = list(cfg.upsampling_ratios)
        cm = list(cfg.channel_multiples)
        if len(ups) != len(cm) or len(ups) > native.MAX_BLOCKS:
            raise ValueError("ace355: unsupported VAE block configuration")
        c = native.VaeConfigC(cfg.decoder_channels, cfg.decoder_input_channels, cfg.audio_channels, len(ups),
                              (C.c_int32 * native.MAX_BLOCKS)(*cm), (C.c_int32 * native.MAX_BLOCKS)(*ups))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_vae_create(C.byref(c), C.byref(h)), "vae_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):
                self._lib.ace355_vae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def hop(self) -> int:
        return int(self._lib.ace355_vae_hop(self._h))

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Keys of ``AutoencoderOobleck.state_dict()``: the decoder half is required, the encoder half optional
        (``encode`` needs it)."""
        self.has_encoder = any(k.startswith("encoder.") for k in sd)
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if not (name.startswith("decoder.") or name.startswith("encoder.")):
                    continue
                t = t.detach()
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                t = t.contiguous()
                dt = native.DTYPE_F32 if t.dtype == torch.float32 else native.DTYPE_BF16
                native.check(self._lib.ace355_vae_load_tensor(self._h, name.encode(), native.ptr(t), dt, t.numel(),
                                                              1 if t.is_cuda else 0), f"vae_load_tensor({name})")
            native.check(self._lib.ace355_vae_finalize(self._h), "vae_finalize")

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, 64, T] (the reference's layout) -> waveform fp32 [B, 2, hop*T] on this device."""
        B, Cc, T = z.shape
        if Cc != self.cfg.decoder_input_channels:
            raise ValueError(f"ace355: expected {self.cfg.decoder_input_channels} latent channels, got {Cc}")
        z = z.detach().to(self.device, torch.float32).contiguous()
        out = torch.empty(B, self.cfg.audio_channels, self.hop * T, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_vae_decode(self._h, native.ptr(z), B, T, native.ptr(out), native.current_stream_ptr()),
                         "vae_decode")
        return out

    def set_decode_budget(self, budget_bytes: int = 0, overlap_frames: int = 0) -> None:
        """Activation budget of ``decode`` (bytes; the reference sizes its chunks by free VRAM, handler/memory_utils.py:48-83)
        and the halo per window side in latent frames (the reference's ``overlap``, handler/vae_decode.py:16); 0 = unchanged."""
        native.check(self._lib.ace355_vae_set_decode_budget(self._h, int(budget_bytes), int(overlap_frames)), "vae_set_decode_budget")

    def decode_plan(self, B: int, T: int) -> Dict[str, int]:
        """How ``decode`` would split (B, T): items per window, core frames per window (== T: whole sequence), halo, bytes."""
        nb, tc, ov, by = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        native.check(self._lib.ace355_vae_decode_plan(self._h, B, T, C.byref(nb), C.byref(tc), C.byref(ov), C.byref(by)), "vae_decode_plan")
        return {"items_per_window": nb.value, "core_frames": tc.value, "overlap_frames": ov.value, "activation_bytes": by.value}

    def encode(self, audio: torch.Tensor, noise: torch.Tensor = None, generator: torch.Generator = None, sample: bool = True) -> torch.Tensor:
        """audio [B, 2, L] -> latents fp32 [B, 64, T] = ``vae.encode(audio).latent_dist.sample()`` (handler/vae_encode.py:66).

        ``noise`` [B, 64, T] makes the draw reproducible (parity tests); otherwise it is drawn here with ``generator``;
        ``sample=False`` returns the mean (``latent_dist.mode()``)."""
        B, Cc, L = audio.shape
        if Cc != self.cfg.audio_channels:
            raise ValueError(f"ace355: expected {self.cfg.audio_channels} audio channels, got {Cc}")
        T = int(self._lib.ace355_vae_latent_frames(self._h, L))
        if T <= 0:
            raise RuntimeError("ace355: the VAE encoder is not available (encoder.* weights not loaded) or the audio is too short")
        audio = audio.detach().to(self.device, torch.float32).contiguous()
        Z = self.cfg.decoder_input_channels
        if sample and noise is None:
            noise = torch.randn(B, Z, T, device=self.device, dtype=torch.float32, generator=generator)
        if noise is not None:
            if tuple(noise.shape) != (B, Z, T):
                raise ValueError(f"ace355: noise must be [{B}, {Z}, {T}]")
            noise = noise.detach().to(self.device, torch.float32).contiguous()
        out = torch.empty(B, Z, T, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_vae_encode(self._h, native.ptr(audio), native.ptr(noise) if sample else 0, B, L, native.ptr(out),
                                                     native.current_stream_ptr()), "vae_encode")
        return out

    def set_profile(self, enable: bool) -> None:
        native.check(self._lib.ace355_vae_set_profile(self._h, 1 if enable else 0), "vae_set_profile")

    def get_profile(self) -> Dict[str, float]:
        ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
        native.check(self._lib.ace355_vae_get_profile(self._h, C.byref(ms), C.byref(fl), C.byref(n)), "vae_get_profile")
        return {"conv_ms": ms.value, "conv_flops": fl.value, "conv_launches": n.value}


def peak_normalize(wav: torch.Tensor) -> torch.Tensor:
    """handler/generate_music_decode.py:191-195, in place on the device."""
    lib = native.lib()
    assert wav.is_cuda and wav.dtype == torch.float32 and wav.is_contiguous()
    with torch.cuda.device(wav.device):
        native.check(lib.ace355_peak_normalize(native.ptr(wav), wav.shape[0], wav[0].numel(), native.current_stream_ptr()),
                     "peak_normalize")
    return wav


def latent_check(lat: torch.Tensor):
    """handler/generate_music_decode.py:66-77 -> (has_nan_or_inf, all_zero)."""
    lib = native.lib()
    lat = lat.detach().float().contiguous()
    flags = (C.c_int32 * 2)()
    with torch.cuda.device(lat.device):
        native.check(lib.ace355_latent_check(native.ptr(lat), lat.numel(), flags, native.current_stream_ptr()), "latent_check")
    return bool(flags[0]), bool(flags[1])
