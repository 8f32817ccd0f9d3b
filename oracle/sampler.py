"""fp32 CPU restatement of the base/sft flow-matching sampler (test oracle).

Reference: AceStepConditionGenerationModel.generate_audio,
/root/reference/acestep/models/base/modeling_acestep_v15_base.py:1783-1989 (``base.py``)
and the sft twin's explicit ``timesteps=`` (sft/modeling_acestep_v15_base.py:1864-1875).

The oracle starts where the drop-in boundary starts: conditions are
already prepared (``encoder_hidden_states``, ``context_latents``), exactly what
``_mlx_run_diffusion`` / ``_native_run_diffusion`` receive.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import torch

from . import apg as _apg
from .dit import CrossCache, DitConfig, dit_forward

Tensor = torch.Tensor


def schedule(infer_steps: int, shift: float = 1.0, timesteps: Optional[Sequence[float]] = None, dtype=torch.float32) -> Tensor:
    """base.py:1864-1867 (+ sft :1864-1875): linspace(1,0,steps+1), optional shift warp."""
    if timesteps is not None:
        return torch.as_tensor(timesteps, dtype=dtype)
    t = torch.linspace(1.0, 0.0, infer_steps + 1, dtype=dtype)
    if shift != 1.0:
        t = shift * t / (1 + (shift - 1) * t)
    return t


def prepare_noise(shape, seed: Union[int, List[int], None], dtype=torch.float32) -> Tensor:
    """base.py:1733-1770 on the CPU device: per-item torch.Generator when seed is a list."""
    bsz, T, C = shape
    if seed is None:
        return torch.randn(shape, dtype=dtype)
    if isinstance(seed, list):
        out = []
        for s in seed:
            if s is None or s < 0:
                out.append(torch.randn(1, T, C, dtype=dtype))
            else:
                g = torch.Generator(device="cpu").manual_seed(int(s))
                out.append(torch.randn(1, T, C, generator=g, dtype=dtype))
        return torch.cat(out, dim=0)
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    return torch.randn(shape, generator=g, dtype=dtype)


def generate_audio(
    cfg: DitConfig,
    w: Dict[str, Tensor],
    null_condition_emb: Tensor,
    encoder_hidden_states: Tensor,
    context_latents: Tensor,
    seed: Union[int, List[int], None] = None,
    infer_method: str = "ode",
    infer_steps: int = 30,
    diffusion_guidance_sale: float = 7.0,
    cfg_interval_start: float = 0.0,
    cfg_interval_end: float = 1.0,
    use_adg: bool = False,
    shift: float = 1.0,
    timesteps: Optional[Sequence[float]] = None,
    audio_cover_strength: float = 1.0,
    cover_noise_strength: float = 0.0,
    src_latents: Optional[Tensor] = None,
    encoder_hidden_states_non_cover: Optional[Tensor] = None,
    context_latents_non_cover: Optional[Tensor] = None,
    noise: Optional[Tensor] = None,
    trace: Optional[list] = None,
    sde_noise: Optional[Tensor] = None,
    sde_next_from_schedule: bool = False,
) -> Tensor:
    """The sampling loop of base.py:1861-1989 given prepared conditions; returns target_latents [B,T,64].

    ``sde_noise`` [steps,B,T,64] replaces the unseeded per-step ``randn_like`` draws of the "sde" branch (so that it can be
    replayed); ``sde_next_from_schedule`` selects the TURBO model's renoise level ``t_schedule[step_idx + 1]``
    (turbo.py:1980-1984) instead of the base model's ``1 - (step_idx + 1) / infer_steps`` (base.py:1972)."""
    dtype = context_latents.dtype
    bsz = context_latents.shape[0]
    t = schedule(infer_steps, shift, timesteps, dtype)
    if timesteps is not None:
        infer_steps = len(t) - 1
    cover_steps = int(infer_steps * audio_cover_strength)
    if noise is None:
        noise = prepare_noise((bsz, context_latents.shape[1], context_latents.shape[-1] // 2), seed, dtype)
    cache = CrossCache()
    momentum = _apg.MomentumBuffer()

    if cover_noise_strength > 0.0:  # base.py:1879-1900
        eff = 1.0 - cover_noise_strength
        t_values = t[:-1].tolist()
        nearest = min(t_values, key=lambda x: abs(x - eff))
        start_idx = t_values.index(nearest)
        xt = nearest * noise + (1 - nearest) * src_latents
        t = t[start_idx:]
        infer_steps = len(t) - 1
        cover_steps = int(infer_steps * audio_cover_strength)
    else:
        xt = noise

    do_cfg = diffusion_guidance_sale > 1.0
    enc, ctx = encoder_hidden_states, context_latents
    if do_cfg:  # base.py:1905-1911
        enc = torch.cat([enc, null_condition_emb.expand_as(enc)], dim=0)
        ctx = torch.cat([ctx, ctx], dim=0)

    switched = False
    for step_idx, (t_curr, t_prev) in enumerate(zip(t[:-1], t[1:])):
        if step_idx >= cover_steps and not switched:  # base.py:1916-1927
            switched = True
            enc_nc, ctx_nc = encoder_hidden_states_non_cover, context_latents_non_cover
            if do_cfg:
                enc_nc = torch.cat([enc_nc, null_condition_emb.expand_as(enc_nc)], dim=0)
                ctx_nc = torch.cat([ctx_nc, ctx_nc], dim=0)
            enc, ctx = enc_nc, ctx_nc
            cache = CrossCache()
        x = torch.cat([xt, xt], dim=0) if do_cfg else xt
        tt = t_curr * torch.ones((x.shape[0],), dtype=dtype)
        vt = dit_forward(cfg, w, x, tt, tt, enc, ctx, cache)
        apply_cfg = bool(t_curr >= cfg_interval_start and t_curr <= cfg_interval_end)
        if do_cfg:  # base.py:1946-1966
            pc, pu = vt.chunk(2)
            if apply_cfg:
                if not use_adg:
                    vt = _apg.apg_forward(pc, pu, diffusion_guidance_sale, momentum, dims=[1])
                else:
                    vt = _apg.adg_forward(xt, pc, pu, t_curr, diffusion_guidance_sale)
            else:
                vt = pc
        if infer_method == "sde":  # base.py:1968-1973 (unseeded renoise: not reproducible)
            tb = t_curr * torch.ones((bsz,), dtype=dtype)
            clean = xt - vt * tb[:, None, None]
            nt = float(t_prev) if sde_next_from_schedule else 1.0 - (float(step_idx + 1) / infer_steps)
            eps = torch.randn_like(clean) if sde_noise is None else sde_noise[step_idx].to(dtype)
            xt = nt * eps + (1 - nt) * clean
        elif infer_method == "ode":  # base.py:1974-1979
            dt = t_curr - t_prev
            xt = xt - vt * (dt * torch.ones((bsz,), dtype=dtype))[:, None, None]
        if trace is not None:
            trace.append(xt.clone())
    return xt


# --------------------------------------------------------------------------- turbo (8-step distilled, no CFG)
TURBO_VALID_SHIFTS = [1.0, 2.0, 3.0]
TURBO_VALID_TIMESTEPS = [
    1.0, 0.9545454545454546, 0.9333333333333333, 0.9, 0.875, 0.8571428571428571, 0.8333333333333334, 0.7692307692307693, 0.75,
    0.6666666666666666, 0.6428571428571429, 0.625, 0.5454545454545454, 0.5, 0.4, 0.375, 0.3, 0.25, 0.2222222222222222, 0.125]
TURBO_SHIFT_TIMESTEPS = {
    1.0: [1.0, 0.875, 0.75, 0.625, 0.5, 0.375, 0.25, 0.125],
    2.0: [1.0, 0.9333333333333333, 0.8571428571428571, 0.7692307692307693, 0.6666666666666666, 0.5454545454545454, 0.4, 0.2222222222222222],
    3.0: [1.0, 0.9545454545454546, 0.9, 0.8333333333333334, 0.75, 0.6428571428571429, 0.5, 0.3],
}


def turbo_schedule(shift: float = 3.0, timesteps: Optional[Sequence[float]] = None) -> List[float]:
    """turbo/modeling_acestep_v15_turbo.py:1807-1865: fixed 8-value tables per shift in {1,2,3}; explicit ``timesteps`` lose
    trailing zeros, are truncated to 20 and snapped to the 20 trained values (an empty list falls back to the shift table)."""
    if timesteps is not None:
        ts = [float(x) for x in (timesteps.tolist() if isinstance(timesteps, torch.Tensor) else list(timesteps))]
        while ts and ts[-1] == 0:
            ts.pop()
        if len(ts) >= 1:
            ts = ts[:20]
            return [min(TURBO_VALID_TIMESTEPS, key=lambda x: abs(x - t)) for t in ts]
    return list(TURBO_SHIFT_TIMESTEPS[min(TURBO_VALID_SHIFTS, key=lambda x: abs(x - shift))])


def generate_audio_turbo(cfg: DitConfig, w: Dict[str, Tensor], encoder_hidden_states: Tensor, context_latents: Tensor,
                         seed: Union[int, List[int], None] = None, shift: float = 3.0, timesteps: Optional[Sequence[float]] = None,
                         infer_method: str = "ode", audio_cover_strength: float = 1.0, cover_noise_strength: float = 0.0,
                         src_latents: Optional[Tensor] = None, encoder_hidden_states_non_cover: Optional[Tensor] = None,
                         context_latents_non_cover: Optional[Tensor] = None, noise: Optional[Tensor] = None,
                         sde_noise: Optional[Tensor] = None) -> Tensor:
    """Sampling loop of the turbo model (turbo.py:1780-1995) given prepared conditions: no CFG / null branch, the last step is
    ``x0 = xt - vt * t`` (:1976-1978), i.e. the base loop on ``table + [0]`` with guidance 1 (no momentum, no interval).
    "sde" renoises to the NEXT TABLE VALUE (turbo.py:1980-1984), not to the base model's linear level; the last step's level
    is the appended 0, which reproduces ``x0 = xt - vt * t`` for both methods."""
    table = turbo_schedule(shift, timesteps)
    return generate_audio(cfg, w, torch.zeros(1, 1, cfg.hidden_size), encoder_hidden_states, context_latents, seed=seed,
                          infer_method=infer_method, infer_steps=len(table), diffusion_guidance_sale=1.0, shift=shift,
                          timesteps=table + [0.0], audio_cover_strength=audio_cover_strength, cover_noise_strength=cover_noise_strength,
                          src_latents=src_latents, encoder_hidden_states_non_cover=encoder_hidden_states_non_cover,
                          context_latents_non_cover=context_latents_non_cover, noise=noise, sde_noise=sde_noise,
                          sde_next_from_schedule=True)
