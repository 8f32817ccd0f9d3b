"""Host-side mirror of the reference's output stage on the native library.

Reference: acestep/audio_utils.py:24-62 (``normalize_audio``), :65-313 (``AudioSaver``: ``save_audio``,
``convert_audio``, ``save_batch``) and the per-item loop of acestep/inference.py:649-726.  Same names, argument meaning
and error behaviour; the work moves as follows:

* ``normalize_audio`` on a GPU tensor is two kernels (absmax, gain) instead of three host passes per item;
  ``normalize_audio_batch`` does a whole decoded batch ``[B, C, S]`` in one call, one peak per item as the reference's
  per-item loop computes it.
* ``AudioSaver.save_batch`` on a GPU batch interleaves and quantises on the GPU, copies the converted samples once and
  encodes all (item, block) jobs on one pool of host threads (FLAC frames are independent).
* "flac" (PCM_16), "wav" and "wav32" (IEEE float32, what the reference's torchaudio/soundfile path writes for a float32
  tensor) are native.  "mp3" / "opus" / "aac" need ffmpeg in the reference (audio_utils.py:149-157) and raise here, so a
  host integrating this keeps its own path for them.

No CPU fallback: CPU tensors are staged through the GPU path only when a GPU is present; the host codecs (FLAC/WAV
encode, FLAC decode) are plain host functions of the same library and fail loudly if it is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import native

FORMATS = ["flac", "wav", "mp3", "wav32", "opus", "aac"]
_NATIVE = {"flac": 0, "wav": 1, "wav32": 1}  # ACE355_AUDIO_*
_EXTS = [".flac", ".wav", ".mp3", ".opus", ".aac", ".m4a"]


def normalize_audio(audio_data, target_db: float = -1.0):
    """audio_utils.py:24-62.  GPU tensors: native kernels, returns a new tensor (the reference clones); silent input
    (peak < 1e-6) is returned as is.  Other inputs (CPU tensors, numpy) are not this backend's business and raise."""
    if not (isinstance(audio_data, torch.Tensor) and audio_data.is_cuda):
        raise TypeError("ace355.normalize_audio handles CUDA tensors; keep the host implementation for CPU / numpy inputs")
    if audio_data.numel() == 0:
        return audio_data
    out = audio_data.detach().to(torch.float32).clone(memory_format=torch.contiguous_format)
    peak = (C.c_float * 1)()
    with torch.cuda.device(out.device):
        native.check(native.lib().ace355_normalize_audio(native.ptr(out), 1, out.numel(), float(target_db), peak, native.current_stream_ptr()),
                     "normalize_audio")
    return audio_data if peak[0] < 1e-6 else out


def normalize_audio_batch(wav: torch.Tensor, target_db: float = -1.0, inplace: bool = False):
    """The normalisation of inference.py:673-688 for a whole decoded batch [B, C, S] on the GPU: item b gets
    ``normalize_audio(wav[b], target_db)``.  Returns (normalised batch, per-item peaks before scaling)."""
    assert wav.is_cuda and wav.dim() == 3
    out = wav if (inplace and wav.dtype == torch.float32 and wav.is_contiguous()) else wav.detach().to(torch.float32).clone(memory_format=torch.contiguous_format)
    B = out.shape[0]
    peaks = (C.c_float * B)()
    with torch.cuda.device(out.device):
        native.check(native.lib().ace355_normalize_audio(native.ptr(out), B, out[0].numel(), float(target_db), peaks, native.current_stream_ptr()),
                     "normalize_audio")
    return out, torch.tensor(list(peaks), dtype=torch.float32)


# ---------------------------------------------------------------------------------------------------- host codecs
def _as_np(a, dtype) -> np.ndarray:
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    return np.ascontiguousarray(a, dtype=dtype)


def flac_encode_pcm16(pcm, sample_rate: int, n_threads: int = 0) -> bytes:
    """pcm: int16 [frames, channels] (interleaved) -> FLAC stream."""
    pcm = _as_np(pcm, np.int16)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    frames, ch = pcm.shape
    lib = native.lib()
    cap = int(lib.ace355_flac_bound(frames, ch))
    buf = (C.c_uint8 * max(cap, 64))()
    n = C.c_int64(0)
    native.check(lib.ace355_flac_encode_pcm16(pcm.ctypes.data, frames, ch, int(sample_rate), int(n_threads), buf, cap, C.byref(n)), "flac_encode_pcm16")
    return bytes(memoryview(buf)[:n.value])


def flac_decode_pcm16(data: bytes, verify_md5: bool = True):
    """FLAC stream (<= 16 bits per sample) -> (int16 [frames, channels], sample_rate)."""
    lib = native.lib()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    frames, ch, sr, bps = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    native.check(lib.ace355_flac_info(buf, len(data), C.byref(frames), C.byref(ch), C.byref(sr), C.byref(bps)), "flac_info")
    out = np.zeros((frames.value, ch.value), dtype=np.int16)
    native.check(lib.ace355_flac_decode_pcm16(buf, len(data), out.ctypes.data, out.size, 1 if verify_md5 else 0), "flac_decode_pcm16")
    return out, int(sr.value)


def wav_encode(samples, sample_rate: int) -> bytes:
    """samples: float32 or int16 [frames, channels] (interleaved) -> RIFF/WAVE (IEEE float32 or PCM_16)."""
    a = samples.detach().cpu().numpy() if isinstance(samples, torch.Tensor) else np.asarray(samples)
    is_float = a.dtype != np.int16
    a = np.ascontiguousarray(a, dtype=np.float32 if is_float else np.int16)
    if a.ndim == 1:
        a = a[:, None]
    frames, ch = a.shape
    lib = native.lib()
    cap = int(lib.ace355_wav_bound(frames, ch, int(is_float)))
    buf = (C.c_uint8 * cap)()
    n = C.c_int64(0)
    native.check(lib.ace355_wav_encode(a.ctypes.data, frames, ch, int(sample_rate), int(is_float), buf, cap, C.byref(n)), "wav_encode")
    return bytes(memoryview(buf)[:n.value])


def float_to_pcm16(wav: torch.Tensor) -> torch.Tensor:
    """GPU: [B, C, S] (or [C, S]) float32 -> interleaved int16 [B, S, C] (or [S, C]); lrintf(x * 32767), saturated."""
    assert wav.is_cuda
    w = wav.detach().to(torch.float32).contiguous()
    squeeze = w.dim() == 2
    if squeeze:
        w = w[None]
    B, Cn, S = w.shape
    out = torch.empty(B, S, Cn, dtype=torch.int16, device=w.device)
    with torch.cuda.device(w.device):
        native.check(native.lib().ace355_audio_interleave(native.ptr(w), B, Cn, S, native.ptr(out), 1, native.current_stream_ptr()), "audio_interleave")
    return out[0] if squeeze else out


class AudioSaver:
    """AudioSaver of acestep/audio_utils.py:65-313 for GPU tensors and the PCM container formats."""

    def __init__(self, default_format: str = "flac", n_threads: int = 0):
        self.default_format = default_format.lower()
        if self.default_format not in FORMATS:  # audio_utils.py:75-78: unknown -> flac
            self.default_format = "flac"
        self.n_threads = int(n_threads)

    def _resolve(self, output_path, format: Optional[str]):
        fmt = (format or self.default_format).lower()
        if fmt not in FORMATS:
            fmt = self.default_format
        path = Path(output_path)
        ext = ".wav" if fmt == "wav32" else f".{fmt}"
        if path.suffix.lower() not in _EXTS:  # audio_utils.py:110-117
            path = path.with_suffix(ext)
        if fmt not in _NATIVE:
            raise NotImplementedError(f"ace355: '{fmt}' is encoded by ffmpeg in the reference (audio_utils.py:149-157); keep the host path")
        return fmt, path

    @staticmethod
    def _to_batch(audio, channels_first: bool) -> torch.Tensor:
        t = torch.from_numpy(audio).float() if isinstance(audio, np.ndarray) else audio.detach().float()
        if t.dim() == 1:
            t = t[None]
        if not channels_first and t.dim() == 2 and t.shape[0] > t.shape[1]:  # the reference's [samples, channels] heuristic
            t = t.T
        return t

    def save_audio(self, audio_data, output_path, sample_rate: int = 48000, format: Optional[str] = None, channels_first: bool = True) -> str:
        """One item [channels, samples] -> file; returns the path actually written."""
        fmt, path = self._resolve(output_path, format)
        t = self._to_batch(audio_data, channels_first)
        self._save([t], [path], sample_rate, fmt)
        return str(path)

    def save_batch(self, audio_batch, output_dir, file_prefix: str = "audio", sample_rate: int = 48000, format: Optional[str] = None,
                   channels_first: bool = True) -> List[str]:
        """audio_utils.py:259-313: files ``{prefix}_{i:04d}.{ext}`` under output_dir."""
        out_dir = Path(output_dir)
        out_dir.mkdir(parents=True, exist_ok=True)
        if isinstance(audio_batch, torch.Tensor) and audio_batch.dim() == 3:
            items = [audio_batch[i] for i in range(audio_batch.shape[0])]
            whole = audio_batch if channels_first else None
        elif isinstance(audio_batch, list):
            items, whole = audio_batch, None
        else:
            items, whole = [audio_batch], None
        fmt = None
        paths = []
        for i in range(len(items)):
            fmt, p = self._resolve(out_dir / f"{file_prefix}_{i:04d}", format)
            paths.append(p)
        if whole is not None and whole.is_cuda:
            self._save_device_batch(whole, paths, sample_rate, fmt)
        else:
            self._save([self._to_batch(a, channels_first) for a in items], paths, sample_rate, fmt)
        return [str(p) for p in paths]

    def save_paths(self, audio_batch: torch.Tensor, paths: Sequence[Union[str, Path]], sample_rate: int = 48000, format: Optional[str] = None) -> List[str]:
        """The loop of inference.py:700-716 in one call: item i of a [B, C, S] batch -> paths[i] (names are the
        caller's, e.g. ``{uuid}.flac``)."""
        resolved = [self._resolve(p, format) for p in paths]
        fmt = resolved[0][0]
        ps = [p for _, p in resolved]
        if audio_batch.is_cuda:
            self._save_device_batch(audio_batch, ps, sample_rate, fmt)
        else:
            self._save([audio_batch[i] for i in range(audio_batch.shape[0])], ps, sample_rate, fmt)
        return [str(p) for p in ps]

    def _save_device_batch(self, wav: torch.Tensor, paths, sample_rate: int, fmt: str):
        w = wav.detach().to(torch.float32).contiguous()
        B, Cn, S = w.shape
        arr = (C.c_char_p * B)(*[os.fsencode(str(p)) for p in paths])
        with torch.cuda.device(w.device):
            native.check(native.lib().ace355_save_audio_batch(native.ptr(w), B, Cn, S, int(sample_rate), _NATIVE[fmt], arr, self.n_threads,
                                                              native.current_stream_ptr()), "save_audio_batch")

    def _save(self, items, paths, sample_rate: int, fmt: str):
        # the float -> PCM_16 / interleave step is a GPU kernel: host tensors are staged through the device
        if not torch.cuda.is_available():
            raise RuntimeError("ace355: the output stage runs on the GPU and no GPU is visible (no CPU fallback)")
        dev = next((t.device for t in items if isinstance(t, torch.Tensor) and t.is_cuda), torch.device("cuda:0"))
        items = [t.to(dev) for t in items]
        if items and all(t.shape == items[0].shape for t in items):
            return self._save_device_batch(torch.stack(items), paths, sample_rate, fmt)
        for t, p in zip(items, paths):
            self._save_device_batch(t[None], [p], sample_rate, fmt)

    def convert_audio(self, input_path, output_path, output_format: str, remove_input: bool = False) -> str:
        """audio_utils.py:217-257 for flac -> {wav, wav32, flac} (the decoder handles <= 16-bit FLAC)."""
        input_path = Path(input_path)
        if not input_path.exists():
            raise FileNotFoundError(f"Input file not found: {input_path}")
        if input_path.suffix.lower() != ".flac":
            raise NotImplementedError("ace355.convert_audio reads FLAC; keep the host path for other inputs")
        pcm, sr = flac_decode_pcm16(input_path.read_bytes())
        fmt, path = self._resolve(output_path, output_format)
        if fmt == "flac":
            data = flac_encode_pcm16(pcm, sr, self.n_threads)
        else:  # torchaudio.load returns float32 = int16 / 32768 (libsndfile's PCM -> float rule)
            data = wav_encode((pcm.astype(np.float32) / np.float32(32768.0)), sr)
        path.write_bytes(data)
        if remove_input:
            input_path.unlink()
        return str(path)
