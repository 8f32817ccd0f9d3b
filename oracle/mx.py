"""fp32 / integer restatement of OCP MXFP8 block quantisation (test oracle for csrc mx_quant_kernel and the MX GEMM).

The reference's fp8 knob is torchao (`fp8_weight_only` / `w8a8_dynamic` on the DiT Linears,
/root/reference/acestep/core/generation/handler/init_service_loader.py:89-113); torchao is absent from /root/reference and from the
image, so nothing here can be pinned against it: **parity unpinned**.  What is restated is the published OCP Microscaling
Formats (MX) v1.0 specification, section 6.3 (the format gfx950's v_mfma_scale_* instructions consume): blocks of 32 consecutive
elements along K share one E8M0 scale X = 2^(floor(log2(max|v|)) - emax_elem), emax_elem = 8 for e4m3; elements are
round-to-nearest-even(v / X) saturated to the e4m3 range (+-448).
"""
from __future__ import annotations

from typing import Tuple

import torch

BLOCK = 32


def mx_quantize(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [M, K] (any float dtype; values are taken as fp32) -> (q float8_e4m3fn [M, K], scale_exp uint8 [M, K/32] biased by 127)."""
    M, K = x.shape
    assert K % BLOCK == 0
    v = x.detach().to(torch.float32).reshape(M, K // BLOCK, BLOCK)
    amax = v.abs().amax(dim=-1)
    # floor(log2(amax)) from the exponent field, exactly as the kernel does (amax == 0 -> field 0)
    ex = (amax.view(torch.int32) >> 23) & 0xFF
    sb = (ex - 8).clamp(0, 254)
    inv = ((254 - sb) << 23).to(torch.int32).view(torch.float32)  # 2^-(sb - 127)
    q = (v * inv[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.reshape(M, K), sb.to(torch.uint8)


def mx_dequantize(q: torch.Tensor, sb: torch.Tensor) -> torch.Tensor:
    M, K = q.shape
    scale = torch.pow(2.0, sb.to(torch.float32) - 127.0)
    return (q.to(torch.float32).reshape(M, K // BLOCK, BLOCK) * scale[..., None]).reshape(M, K)


def pack_scales(sb: torch.Tensor, rows_pad: int) -> torch.Tensor:
    """[M, K/32] uint8 -> the library's layout uint32 [K/128][rows_pad]: byte b of word [kt][row] = block 4 kt + b."""
    M, nb = sb.shape
    assert nb % 4 == 0
    w = sb.to(torch.int64).reshape(M, nb // 4, 4)
    word = w[..., 0] | (w[..., 1] << 8) | (w[..., 2] << 16) | (w[..., 3] << 24)
    out = torch.zeros(nb // 4, rows_pad, dtype=torch.int64)
    out[:, :M] = word.t()
    return out


def fp8_weight_only_roundtrip(weights: dict) -> dict:
    """The numerics of the reference's `quantization="fp8_weight_only"` (torchao Float8WeightOnlyConfig on every nn.Linear of the DiT
    outside tokenizer / detokenizer, init_service_loader.py:95-113; torchao absent: **parity unpinned**, restated from its documented
    scheme): per output channel scale s = max(amax, 1e-12) / 448 in fp32, q = e4m3_rne(w / s), dequantised weight = bf16(q * s).
    Returns a copy of a decoder state dict with every Linear weight replaced; Conv1d / ConvTranspose1d (proj_in / proj_out), norms,
    biases and the scale-shift tables are not Linears and stay."""
    out = {}
    for k, v in weights.items():
        is_linear = v.dim() == 2 and k.endswith(".weight") and not k.startswith("proj_in") and not k.startswith("proj_out") and "norm" not in k
        if not is_linear:
            out[k] = v
            continue
        w = v.detach().to(torch.bfloat16).to(torch.float32)
        s = w.abs().amax(dim=1, keepdim=True).clamp_min(1e-12) / 448.0
        q = (w / s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
        out[k] = (q * s).to(torch.bfloat16).to(torch.float32)
    return out
