"""PyTorch custom ops (``torch.library``) over the C ABI: ``torch.ops.ace355.*``.

north_star asks that the native path "drops in behind the Gradio/API front-ends via PyTorch-ROCm custom ops"; the
reference's seam (handler/service_generate_execute.py:144-194, handler/vae_decode.py:39-48) calls Python mixin methods,
which ``backend.py`` provides.  These ops are the same entry points in op form, for callers that want dispatcher-visible
operators (``torch.compile`` / ``torch.export`` graphs, FakeTensor shape propagation, profiler op names):

  ace355::dit_forward   AceStepDiTModel.forward            (modeling_acestep_v15_base.py:1303-1507)
  ace355::dit_sample    the loop of generate_audio         (modeling_acestep_v15_base.py:1913-1981)
  ace355::vae_decode    AutoencoderOobleck.decode(z).sample (handler/vae_decode_chunks.py:42,95)
  ace355::vae_encode    vae.encode(x).latent_dist.sample()  (handler/vae_encode.py:66)
  ace355::peak_normalize  post-decode peak clip             (handler/generate_music_decode.py:191-195)

A native handle (``NativeDit`` / ``NativeVae``: weights packed in HBM, condition slots) is opaque state, not a tensor: it is
registered once with ``register_handle`` and travels through the op as an integer key.  The ops have CUDA(=ROCm)
implementations only - there is no CPU kernel; fake (meta) implementations give output shapes without touching a GPU.
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Optional

import torch
from torch import Tensor

_HANDLES: Dict[int, object] = {}
_next = itertools.count(1)


def register_handle(obj) -> int:
    """Make a ``NativeDit`` / ``NativeVae`` reachable from the ops; returns the key to pass as ``handle``."""
    for k, v in _HANDLES.items():
        if v is obj:
            return k
    k = next(_next)
    _HANDLES[k] = obj
    return k


def release_handle(key: int) -> None:
    _HANDLES.pop(int(key), None)


def _get(key: int):
    try:
        return _HANDLES[int(key)]
    except KeyError:
        raise RuntimeError(f"ace355: unknown native handle {key} (register_handle() it first)") from None


@torch.library.custom_op("ace355::dit_forward", mutates_args=(), device_types="cuda")
def dit_forward(handle: int, x: Tensor, ctx: Tensor, t: List[float], t_r: List[float], slots: List[int]) -> Tensor:
    return _get(handle).forward(x, ctx, t, t_r, slots)


@dit_forward.register_fake
def _(handle, x, ctx, t, t_r, slots):
    return x.new_empty(x.shape, dtype=torch.float32)


@torch.library.custom_op("ace355::dit_sample", mutates_args=(), device_types="cuda")
def dit_sample(handle: int, xt0: Tensor, ctx: Tensor, t_sched: Tensor, guidance_scale: float = 7.0, cfg_interval_start: float = 0.0,
               cfg_interval_end: float = 1.0, sde: bool = False, use_adg: bool = False, cond_slot: int = 0, null_slot: int = 1,
               cover_switch_step: int = -1, non_cover_slot: int = 2, ctx_non_cover: Optional[Tensor] = None,
               sde_noise: Optional[Tensor] = None, cond_slots: Optional[List[int]] = None,
               non_cover_slots: Optional[List[int]] = None, sde_next_from_schedule: bool = False) -> Tensor:
    return _get(handle).sample(xt0, ctx, t_sched, guidance_scale, cfg_interval_start, cfg_interval_end, "sde" if sde else "ode", use_adg,
                               cond_slot=cond_slot, null_slot=null_slot, cover_switch_step=None if cover_switch_step < 0 else cover_switch_step,
                               non_cover_slot=non_cover_slot, ctx_non_cover=ctx_non_cover, sde_noise=sde_noise, cond_slots=cond_slots,
                               non_cover_slots=non_cover_slots, sde_next_from_schedule=sde_next_from_schedule)


@dit_sample.register_fake
def _(handle, xt0, ctx, t_sched, guidance_scale=7.0, cfg_interval_start=0.0, cfg_interval_end=1.0, sde=False, use_adg=False,
      cond_slot=0, null_slot=1, cover_switch_step=-1, non_cover_slot=2, ctx_non_cover=None, sde_noise=None, cond_slots=None,
      non_cover_slots=None, sde_next_from_schedule=False):
    return xt0.new_empty(xt0.shape, dtype=torch.float32)


@torch.library.custom_op("ace355::vae_decode", mutates_args=(), device_types="cuda")
def vae_decode(handle: int, z: Tensor, hop: int, audio_channels: int = 2) -> Tensor:
    """z [B,64,T] -> fp32 waveform [B, audio_channels, hop*T].  ``hop`` / ``audio_channels`` are passed so that the fake
    implementation can size the output without the handle."""
    return _get(handle).decode(z)


@vae_decode.register_fake
def _(handle, z, hop, audio_channels=2):
    return z.new_empty((z.shape[0], audio_channels, hop * z.shape[2]), dtype=torch.float32)


@torch.library.custom_op("ace355::vae_encode", mutates_args=(), device_types="cuda")
def vae_encode(handle: int, audio: Tensor, hop: int, latent_channels: int = 64) -> Tensor:
    return _get(handle).encode(audio)


@vae_encode.register_fake
def _(handle, audio, hop, latent_channels=64):
    return audio.new_empty((audio.shape[0], latent_channels, audio.shape[-1] // hop), dtype=torch.float32)


@torch.library.custom_op("ace355::peak_normalize", mutates_args=(), device_types="cuda")
def peak_normalize(wav: Tensor) -> Tensor:
    from .vae import peak_normalize as _pn
    return _pn(wav.clone().contiguous())


@peak_normalize.register_fake
def _(wav):
    return torch.empty_like(wav)
