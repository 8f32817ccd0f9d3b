"""CPU restatement of the tiled-decode window arithmetic and latent/waveform post-processing (test oracle).

Reference (``H/`` = /root/reference/acestep/core/generation/handler/):
  * H/vae_decode_chunks.py:13-81   _tiled_decode_inner (per-sample recursion, overlap halving, direct path)
  * H/vae_decode_chunks.py:83-112  _tiled_decode_gpu (overlap-discard windows, trim by round(frames*upsample))
  * H/generate_music_decode.py:66-96   NaN/Inf/all-zero guards, latent*rescale + shift
  * H/generate_music_decode.py:191-195 .float(), per-item peak, divide by clamp(peak, min=1)
"""
from __future__ import annotations

import math
from typing import Callable, List, Tuple

import torch

Tensor = torch.Tensor


def effective_overlap(chunk_size: int, overlap: int) -> int:
    """vae_decode_chunks.py:31-38: halve overlap until chunk_size - 2*overlap > 0."""
    eff = overlap
    while chunk_size - 2 * eff <= 0 and eff > 0:
        eff //= 2
    return eff


def windows(latent_frames: int, chunk_size: int, overlap: int) -> List[Tuple[int, int, int, int]]:
    """(win_start, win_end, core_start, core_end) per chunk, vae_decode_chunks.py:51-56,88-93."""
    overlap = effective_overlap(chunk_size, overlap)
    if latent_frames <= chunk_size:
        return [(0, latent_frames, 0, latent_frames)]
    stride = chunk_size - 2 * overlap
    if stride <= 0:
        raise ValueError(f"chunk_size {chunk_size} must be > 2 * overlap {overlap}")
    out = []
    for i in range(math.ceil(latent_frames / stride)):
        cs = i * stride
        ce = min(cs + stride, latent_frames)
        out.append((max(0, cs - overlap), min(latent_frames, ce + overlap), cs, ce))
    return out


def tiled_decode(decode_fn: Callable[[Tensor], Tensor], latents: Tensor, chunk_size: int, overlap: int = 64) -> Tensor:
    """_tiled_decode_inner + _tiled_decode_gpu: latents [B,C,T] -> [B,ch,samples]; per-sample sequential."""
    if latents.shape[0] > 1:
        return torch.cat([tiled_decode(decode_fn, latents[b:b + 1], chunk_size, overlap) for b in range(latents.shape[0])], dim=0)
    T = latents.shape[-1]
    wins = windows(T, chunk_size, overlap)
    if len(wins) == 1 and wins[0][1] - wins[0][0] == T and T <= chunk_size:
        return decode_fn(latents)
    pieces = []
    up = None
    for (ws, we, cs, ce) in wins:
        chunk = latents[:, :, ws:we]
        audio = decode_fn(chunk)
        if up is None:
            up = audio.shape[-1] / chunk.shape[-1]
        trim_start = int(round((cs - ws) * up))
        trim_end = int(round((we - ce) * up))
        end = audio.shape[-1] - trim_end if trim_end > 0 else audio.shape[-1]
        pieces.append(audio[:, :, trim_start:end])
    return torch.cat(pieces, dim=-1)


def validate_and_scale_latents(pred: Tensor, latent_shift: float = 0.0, latent_rescale: float = 1.0) -> Tensor:
    """generate_music_decode.py:66-96."""
    if torch.isnan(pred).any() or torch.isinf(pred).any():
        raise RuntimeError("Generation produced NaN or Inf latents.")
    if pred.numel() > 0 and pred.abs().sum() == 0:
        raise RuntimeError("Generation produced zero latents.")
    if latent_shift != 0.0 or latent_rescale != 1.0:
        pred = pred * latent_rescale + latent_shift
    return pred


def peak_normalize(wav: Tensor) -> Tensor:
    """generate_music_decode.py:191-195."""
    wav = wav.float()
    peak = wav.abs().amax(dim=[1, 2], keepdim=True)
    if torch.any(peak > 1.0):
        wav = wav / peak.clamp(min=1.0)
    return wav
