"""Deterministic synthetic weights for the DiT decoder and the Oobleck decoder.

No checkpoints exist on the build or GPU boxes, so parity and
benchmark runs use seeded random weights at the real architecture sizes.  Each tensor
is drawn from its own ``torch.Generator`` seeded by CRC32(name) ^ seed, so any
subset can be regenerated independently and identically on any box (CPU generator).

Two flavours:
  * ``mode="init"``: the reference's ``_init_weights`` statistics
    (modeling_acestep_v15_base.py:558-574, :472, :1299): Linear/Conv ~ N(0, 0.02^2),
    biases 0, RMSNorm weights 1, scale_shift_table ~ N(0,1)/sqrt(D).
  * ``mode="test"``: same matrices, but biases, norm weights and Snake parameters are
    also randomised so that a kernel which drops one of them fails parity.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(name: str, seed: int) -> torch.Generator:
    return torch.Generator(device="cpu").manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def _randn(name: str, shape, seed: int) -> torch.Tensor:
    return torch.randn(tuple(shape), generator=_gen(name, seed), dtype=torch.float32)


def make_dit_weights(shapes: Dict[str, Tuple[int, ...]], hidden_size: int, seed: int = 0, mode: str = "init",
                     std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Weights for every name in ``shapes`` (the keys of AceStepDiTModel.state_dict())."""
    out = {}
    for name, shape in shapes.items():
        if name.endswith("scale_shift_table"):
            out[name] = _randn(name, shape, seed) / hidden_size ** 0.5
        elif name.endswith("norm.weight") or name.endswith("norm_out.weight"):
            out[name] = torch.ones(shape) if mode == "init" else 1.0 + 0.1 * _randn(name, shape, seed)
        elif name.endswith(".bias"):
            out[name] = torch.zeros(shape) if mode == "init" else 0.02 * _randn(name, shape, seed)
        else:
            out[name] = std * _randn(name, shape, seed)
    return out


def make_null_condition_emb(hidden_size: int, seed: int = 0) -> torch.Tensor:
    """AceStepConditionGenerationModel.null_condition_emb ~ N(0,1), shape [1,1,D]."""
    return _randn("null_condition_emb", (1, 1, hidden_size), seed)


def make_vae_weights(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, mode: str = "init") -> Dict[str, torch.Tensor]:
    """Weights for the Oobleck decoder state dict (weight_g / weight_v / bias / alpha / beta).

    ``weight_v`` ~ N(0, 1/fan_in) so activations stay O(1) through ~40 conv layers;
    ``weight_g`` = gain * ||v|| (see ``_vae_gain``) in "init" mode, additionally jittered in "test".
    """
    out = {}
    for name, shape in shapes.items():
        if name.endswith("weight_v"):
            # Conv1d [out,in,K]: fan_in = in*K.  ConvTranspose1d [in,out,K] (stride K/2): each output
            # sample sums 2 taps x in channels.
            if ".conv_t1." in name:
                fan_in = shape[0] * 2
            else:
                fan_in = shape[1] * shape[2]
            out[name] = _randn(name, shape, seed) / fan_in ** 0.5
    for name, shape in shapes.items():
        if name.endswith("weight_g"):
            v = out[name[:-1] + "v"]
            g = v.reshape(v.shape[0], -1).norm(dim=1).reshape(shape) * _vae_gain(name)
            if mode != "init":
                g = g * (1.0 + 0.1 * _randn(name, shape, seed))
            out[name] = g
        elif name.endswith(".bias"):
            out[name] = torch.zeros(shape) if mode == "init" else 0.05 * _randn(name, shape, seed)
        elif name.endswith(".alpha") or name.endswith(".beta"):
            out[name] = torch.zeros(shape) if mode == "init" else 0.3 * _randn(name, shape, seed)
    return out


def _vae_gain(name: str) -> float:
    """Per-layer gains that keep the random decoder's activations O(1) through ~40 layers (rms 0.4-0.9, waveform
    rms ~0.1): an un-scaled random residual stack grows to rms ~250, where sin(alpha*x) is chaotic and bf16 drift is
    meaningless.  Tuned once on the fp32 oracle ("Synthetic VAE weights")."""
    if ".res_unit" in name and ".conv2." in name:
        return 0.3
    if ".res_unit" in name and ".conv1." in name:
        return 0.7
    if ".conv_t1." in name:
        return 0.6
    if name.startswith("decoder.conv2."):
        return 0.2
    return 1.0


def checksum(weights: Dict[str, torch.Tensor], names: Iterable[str] = None) -> float:
    """Order-independent fp64 checksum (sum of per-tensor sum(|w|)) used to pin fixtures to a generator."""
    names = sorted(weights.keys()) if names is None else sorted(names)
    return float(sum(weights[n].double().abs().sum().item() for n in names))
