"""fp32/fp64 CPU restatement of APG / ADG guidance (test oracle).

Reference: /root/reference/acestep/models/base/apg_guidance.py (``apg.py``).
"""
from __future__ import annotations

import torch

Tensor = torch.Tensor


class MomentumBuffer:
    """apg.py:5-13: running_average <- update + momentum * running_average (momentum -0.75)."""

    def __init__(self, momentum: float = -0.75):
        self.momentum = momentum
        self.running_average = 0

    def update(self, value: Tensor):
        self.running_average = value + self.momentum * self.running_average


def project(v0: Tensor, v1: Tensor, dims) -> tuple:
    """apg.py:16-30: parallel / orthogonal split of v0 w.r.t. v1, computed in fp64."""
    dtype = v0.dtype
    v0d, v1d = v0.double(), v1.double()
    v1d = torch.nn.functional.normalize(v1d, dim=dims)
    par = (v0d * v1d).sum(dim=dims, keepdim=True) * v1d
    return par.to(dtype), (v0d - par).to(dtype)


def apg_forward(pred_cond: Tensor, pred_uncond: Tensor, guidance_scale: float, momentum_buffer: MomentumBuffer = None,
                eta: float = 0.0, norm_threshold: float = 2.5, dims=(1,)) -> Tensor:
    """apg.py:33-56.  The sampler calls it with dims=[1] (the T axis of [B,T,64]), base.py:1950-1956."""
    dims = list(dims)
    diff = pred_cond - pred_uncond
    if momentum_buffer is not None:
        momentum_buffer.update(diff)
        diff = momentum_buffer.running_average
    if norm_threshold > 0:
        ones = torch.ones_like(diff)
        diff_norm = diff.norm(p=2, dim=dims, keepdim=True)
        diff = diff * torch.minimum(ones, norm_threshold / diff_norm)
    par, orth = project(diff, pred_cond, dims)
    return pred_cond + (guidance_scale - 1) * (orth + eta * par)


def adg_forward(latents: Tensor, v_cond: Tensor, v_uncond: Tensor, sigma, guidance_scale: float,
                angle_clip: float = 3.14 / 6, apply_norm: bool = False, apply_clip: bool = True) -> Tensor:
    """apg.py:107-180 (angle-based dynamic guidance on x0-hat, per (item, frame) over C)."""
    n, t, c = v_cond.shape
    if not torch.is_tensor(sigma):
        sigma = torch.tensor(float(sigma), dtype=latents.dtype)
    sigma = sigma.reshape(-1)
    sigma = sigma.view(1, 1, 1).expand(n, 1, 1) if sigma.numel() == 1 else sigma.view(n, 1, 1)
    weight = guidance_scale - 1
    weight = weight * (weight > 0) + 1e-3
    x_text = latents - sigma * v_cond
    x_unc = latents - sigma * v_uncond
    diff = x_text - x_unc

    a = x_text.reshape(-1, c).to(torch.float64)
    b = x_unc.reshape(-1, c).to(torch.float64)
    a = a / torch.linalg.norm(a, dim=1, keepdim=True)
    b = b / torch.linalg.norm(b, dim=1, keepdim=True)
    theta = torch.acos(torch.sum(a * b, dim=1, keepdim=True))
    theta_new = torch.clip(weight * theta, -angle_clip, angle_clip) if apply_clip else weight * theta

    d2 = diff.reshape(n * t, c).float()
    u2 = x_unc.reshape(n * t, c).float()
    proj = (torch.sum(d2 * u2, dim=1, keepdim=True) / (torch.sum(u2 * u2, dim=1, keepdim=True) + 1e-8)) * u2
    perp = (d2 - proj).reshape(n, t, c)

    v_new = torch.cos(theta_new) * x_text
    p_new = perp * torch.sin(theta_new) / torch.sin(theta) * (torch.sin(theta) > 1e-3) + perp * weight * (torch.sin(theta) <= 1e-3)
    x_new = v_new + p_new
    if apply_norm:
        x_new = x_new * torch.linalg.norm(x_text, dim=1, keepdim=True) / torch.linalg.norm(x_new, dim=1, keepdim=True)
    out = (latents - x_new) / sigma
    return out.reshape(n, t, c).to(latents.dtype)
