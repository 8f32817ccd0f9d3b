"""ace355: MI355X-native (gfx950) ACE-Step 1.5 denoise + decode hot path.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI shared library (include/ace355.h)
  native.py    ctypes binding of the C-ABI (fails loudly when the library is missing)
  dit.py       NativeDit: weight packing, condition slots, forward, sampler
  vae.py       NativeVae: weight-norm fusion, decode
  cond.py      NativeCondEncoder: lyric / timbre encoders + sequence packing
  lmhints.py   audio-code parsing, FSQ index decode, NativeDetokenizer
  audio_out.py normalize_audio, AudioSaver: GPU normalise / PCM conversion + threaded FLAC / WAV writers
  backend.py   NativeDitMixin / NativeVaeMixin / NativeHandler: the reference's handler seam
  dist.py      one-process-per-GPU data-parallel runner (RCCL broadcast of conditioning)
  weightgen.py deterministic synthetic weights (no checkpoints exist on the boxes)
"""
from .config import CondConfig, DetokConfig, DitConfig, VaeConfig  # noqa: F401

__all__ = ["CondConfig", "DetokConfig", "DitConfig", "VaeConfig"]
