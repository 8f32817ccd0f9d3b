"""NativeDit: Python owner of an ``ace355_dit`` handle (weights, condition slots, forward, sampler).

Host-side mirror of what the reference's MLX path does around ``mlx_generate_diffusion``
(acestep/core/generation/handler/diffusion.py:18-140, mlx_dit_init.py:9-43): convert weights from the
already-loaded PyTorch module's ``state_dict()``, prepare noise on the host with the CPU generator
(modeling_acestep_v15_base.py:1733-1770), build the timestep schedule (:1864-1867), call the native loop.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from typing import Dict, List, Optional, Sequence, Union

import torch

from . import native
from .config import DitConfig

SLOT_COND, SLOT_NULL, SLOT_NON_COVER = 0, 1, 2


def schedule(infer_steps: int, shift: float = 1.0, timesteps: Optional[Sequence[float]] = None) -> torch.Tensor:
    """fp32 schedule of modeling_acestep_v15_base.py:1864-1867 (+ sft ``timesteps=`` :1864-1875)."""
    if timesteps is not None:
        return torch.as_tensor(timesteps, dtype=torch.float32).cpu()
    t = torch.linspace(1.0, 0.0, infer_steps + 1, dtype=torch.float32)
    if shift != 1.0:
        t = shift * t / (1 + (shift - 1) * t)
    return t


def prepare_noise(shape, seed: Union[int, List[int], None]) -> torch.Tensor:
    """prepare_noise (modeling_acestep_v15_base.py:1733-1770) with the CPU generator in fp32: the reference CPU
    path's stream, so seeds reproduce across the CPU reference and this backend (SURVEY.md 7.2 "Noise parity")."""
    bsz, T, Cc = shape
    if seed is None:
        return torch.randn(shape, dtype=torch.float32)
    if isinstance(seed, (list, tuple)):
        out = []
        for s in seed:
            if s is None or s < 0:
                out.append(torch.randn(1, T, Cc, dtype=torch.float32))
            else:
                g = torch.Generator(device="cpu").manual_seed(int(s))
                out.append(torch.randn(1, T, Cc, generator=g, dtype=torch.float32))
        return torch.cat(out, dim=0)
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    return torch.randn(shape, generator=g, dtype=torch.float32)


class NativeDit:
    def __init__(self, cfg: DitConfig, device: Union[str, torch.device] = "cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        self._lib = native.lib()
        mask = 0
        for i, t in enumerate(cfg.layer_types):
            if t == "sliding_attention":
                mask |= 1 << i
        c = native.DitConfigC(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                              cfg.num_key_value_heads, cfg.head_dim, int(cfg.sliding_window or 0), cfg.patch_size,
                              cfg.in_channels, cfg.audio_acoustic_hidden_dim, cfg.rms_norm_eps, cfg.rope_theta, mask)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_dit_create(C.byref(c), C.byref(h)), "dit_create")
        self._h = h
        self._finalized = False

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):
                self._lib.ace355_dit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Ingest ``AceStepDiTModel.state_dict()`` (fp32 or bf16, CPU or this device). Extra keys raise."""
        with torch.cuda.device(self.device):
            for name, t in sd.items():
                if "rotary_emb" in name:
                    continue
                t = t.detach()
                if t.dtype not in (torch.float32, torch.bfloat16):
                    t = t.float()
                t = t.contiguous()
                is_dev = 1 if t.is_cuda else 0
                dt = native.DTYPE_F32 if t.dtype == torch.float32 else native.DTYPE_BF16
                native.check(self._lib.ace355_dit_load_tensor(self._h, name.encode(), native.ptr(t), dt, t.numel(), is_dev),
                             f"dit_load_tensor({name})")
            native.check(self._lib.ace355_dit_finalize(self._h), "dit_finalize")
        self._finalized = True

    # ------------------------------------------------------------------ conditioning
    def set_condition(self, slot: int, enc: torch.Tensor, L: Optional[int] = None) -> None:
        """enc: [L, D] (or [1, D] broadcast over L keys, e.g. null_condition_emb)."""
        enc = enc.detach().to(self.device, torch.float32).reshape(-1, self.cfg.hidden_size).contiguous()
        rows = enc.shape[0]
        L = rows if L is None else L
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_dit_set_condition(self._h, slot, native.ptr(enc), rows, L, native.current_stream_ptr()),
                         "dit_set_condition")
            # (no host sync: `enc` is consumed by work queued on THIS stream, and torch's caching allocator only hands its block
            #  to later work of the same stream - record_stream covers a caller that allocated it on another one)
            enc.record_stream(torch.cuda.current_stream())

    # ------------------------------------------------------------------ compute
    def forward(self, x: torch.Tensor, ctx: torch.Tensor, t: Sequence[float], t_r: Sequence[float], slots: Sequence[int]) -> torch.Tensor:
        N, T, _ = x.shape
        x = x.detach().to(self.device, torch.float32).contiguous()
        ctx = ctx.detach().to(self.device, torch.float32).contiguous()
        out = torch.empty(N, T, self.cfg.audio_acoustic_hidden_dim, device=self.device, dtype=torch.float32)
        ta = (C.c_float * N)(*[float(v) for v in t])
        tr = (C.c_float * N)(*[float(v) for v in t_r])
        sl = (C.c_int32 * N)(*[int(v) for v in slots])
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_dit_forward(self._h, native.ptr(x), native.ptr(ctx), ta, tr, sl, N, T, native.ptr(out),
                                                      native.current_stream_ptr()), "dit_forward")
        return out

    def sample(self, xt0: torch.Tensor, ctx: torch.Tensor, t_sched: torch.Tensor, guidance_scale: float = 7.0,
               cfg_interval_start: float = 0.0, cfg_interval_end: float = 1.0, infer_method: str = "ode", use_adg: bool = False,
               cond_slot: int = SLOT_COND, null_slot: int = SLOT_NULL, cover_switch_step: Optional[int] = None,
               non_cover_slot: int = SLOT_NON_COVER, ctx_non_cover: Optional[torch.Tensor] = None,
               return_step_ms: bool = False, sde_noise: Optional[torch.Tensor] = None,
               cond_slots: Optional[Sequence[int]] = None, non_cover_slots: Optional[Sequence[int]] = None,
               sde_next_from_schedule: bool = False):
        """``cond_slots`` / ``non_cover_slots`` ([B] each): per-item condition slots inside ONE native call (the reference's
        loop takes B distinct ``encoder_hidden_states`` rows, base.py:1905-1911); None = ``cond_slot`` for every item.
        ``sde_next_from_schedule``: the turbo model's renoise level (next table value) instead of the base model's."""
        if infer_method not in {"ode", "sde"}:
            raise ValueError(f"Unsupported infer_method '{infer_method}'. Expected 'ode' or 'sde'.")
        B, T, _ = xt0.shape
        xt0 = xt0.detach().to(self.device, torch.float32).contiguous()
        ctx = ctx.detach().to(self.device, torch.float32).contiguous()
        ts = t_sched.detach().to("cpu", torch.float32).contiguous()
        steps = ts.numel() - 1
        sched = (C.c_float * (steps + 1))(*ts.tolist())
        if ctx_non_cover is not None:
            ctx_non_cover = ctx_non_cover.detach().to(self.device, torch.float32).contiguous()
        if infer_method == "sde":
            if sde_noise is None:  # the reference's unseeded torch.randn_like on the model device (base.py:1777)
                sde_noise = torch.randn(steps, B, T, xt0.shape[-1], device=self.device, dtype=torch.float32)
            sde_noise = sde_noise.detach().to(self.device, torch.float32).contiguous()
            if tuple(sde_noise.shape) != (steps, B, T, xt0.shape[-1]):
                raise ValueError("sde_noise must be [steps, B, T, 64]")
        else:
            sde_noise = None
        def slot_tab(v):
            if v is None:
                return None
            if len(v) != B:
                raise ValueError("per-item slot tables must have one entry per batch item")
            return (C.c_int32 * B)(*[int(x) for x in v])
        ctab, nctab = slot_tab(cond_slots), slot_tab(non_cover_slots)  # kept alive until the call returns
        p = native.SampleParamsC(steps, C.cast(sched, C.POINTER(C.c_float)), float(guidance_scale), float(cfg_interval_start),
                                 float(cfg_interval_end), 0 if infer_method == "ode" else 1, 1 if use_adg else 0, cond_slot,
                                 null_slot, steps if cover_switch_step is None else int(cover_switch_step), non_cover_slot,
                                 native.ptr(ctx_non_cover), native.ptr(sde_noise),
                                 C.cast(ctab, C.POINTER(C.c_int32)) if ctab is not None else None,
                                 C.cast(nctab, C.POINTER(C.c_int32)) if nctab is not None else None,
                                 1 if sde_next_from_schedule else 0)
        out = torch.empty_like(xt0)
        ms = (C.c_float * steps)() if return_step_ms else None
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_dit_sample(self._h, native.ptr(xt0), native.ptr(ctx), B, T, C.byref(p), native.ptr(out),
                                                     ms, native.current_stream_ptr()), "dit_sample")
        if return_step_ms:
            return out, list(ms)
        return out

    def poll_errors(self) -> None:
        """Raise if a queued call hit an asynchronous device-side condition (synchronises the current stream; include/ace355.h)."""
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_dit_poll_errors(self._h, native.current_stream_ptr()), "dit_poll_errors")

    def trim_slots(self, first_unused: int) -> None:
        """Free the cross-K/V buffers of condition slots >= ``first_unused``."""
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_dit_trim_slots(self._h, int(first_unused)), "dit_trim_slots")

    # ------------------------------------------------------------------ precision of the four big projections
    def set_precision(self, precision: str) -> None:
        """"bf16" (default, the reference GPU path's dtype), "mxfp8" (OCP MXFP8 operands on the scaled fp8 MFMA; BASELINE
        configs[4]) or "fp8_weight_only" (the reference's `quantization="fp8_weight_only"` numerics: every Linear weight rounded
        through per-channel e4m3 once, bf16 kernels; init_service_loader.py:89-113; one way - reload the weights to undo)."""
        code = {"bf16": 0, "mxfp8": 1, "fp8_weight_only": 2}.get(precision)
        if code is None:
            raise ValueError(f"unknown precision '{precision}' (bf16 | mxfp8 | fp8_weight_only)")
        with torch.cuda.device(self.device):
            native.check(self._lib.ace355_dit_set_precision(self._h, code), "dit_set_precision")

    def set_norm_fold(self, enable) -> int:
        """RMSNorms folded into the neighbouring GEMM epilogues of sampler calls (default on; include/ace355.h).
        False / 0: off; True / 1: default (calls with >= 64 token rows); 2: every call the kernels support.  Returns the previous mode."""
        native.check(self._lib.ace355_dit_set_norm_fold(self._h, int(enable)), "dit_set_norm_fold")
        prev, self._fold_mode = getattr(self, "_fold_mode", int(os.environ.get("ACE355_NORM_FOLD", "1"))), int(enable)
        return prev

    def set_dual(self, mode) -> int:
        """Dual-chain sampler (include/ace355.h ace355_dit_set_dual): requests of >= 2 songs as two half-batch samplers on two hardware
        queues.  False / 0: one chain; True / 1 (default): two chains for small requests (<= 2400 token rows in all: 2-3 songs of 30 s); 2: whenever
        the request has >= 2 songs.  Returns the previous mode."""
        native.check(self._lib.ace355_dit_set_dual(self._h, int(mode)), "dit_set_dual")
        prev, self._dual_mode = getattr(self, "_dual_mode", int(os.environ.get("ACE355_DUAL", "1"))), int(mode)
        return prev

    def dual_count(self) -> int:
        n = C.c_int64()
        native.check(self._lib.ace355_dit_dual_count(self._h, C.byref(n)), "dit_dual_count")
        return n.value

    def set_dedup(self, enable: bool) -> None:
        """Layer-0 de-duplication of the two CFG copies of a song (include/ace355.h: `ace355_dit_set_dedup`); on by default."""
        native.check(self._lib.ace355_dit_set_dedup(self._h, 1 if enable else 0), "dit_set_dedup")

    def dedup_count(self) -> int:
        """Forwards whose layer 0 ran its QKV projection + self-attention on the conditional half only (`ace355_dit_dedup_count`)."""
        n = C.c_int64()
        native.check(self._lib.ace355_dit_dedup_count(self._h, C.byref(n)), "dit_dedup_count")
        return n.value

    # ------------------------------------------------------------------ hipGraph replay of the sampler loop
    def set_graph(self, enable: bool) -> None:
        """Capture the sampler's launch sequence once per (shapes, schedule, knobs, slots, stream) and replay it afterwards."""
        native.check(self._lib.ace355_dit_set_graph(self._h, 1 if enable else 0), "dit_set_graph")

    def graph_stats(self):
        c, r = C.c_int64(), C.c_int64()
        native.check(self._lib.ace355_dit_graph_stats(self._h, C.byref(c), C.byref(r)), "dit_graph_stats")
        return {"captures": c.value, "replays": r.value}

    # ------------------------------------------------------------------ debug taps
    def set_tap(self, layer: int, dst: Optional[torch.Tensor]) -> None:
        """Copy the fp32 residual stream after decoder layer ``layer`` of every following forward into ``dst``
        ([N*S, hidden] fp32 on this device; None clears).  The caller keeps ``dst`` alive."""
        if dst is not None:
            assert dst.is_cuda and dst.dtype == torch.float32 and dst.is_contiguous()
        native.check(self._lib.ace355_dit_set_tap(self._h, int(layer), native.ptr(dst)), "dit_set_tap")

    # ------------------------------------------------------------------ profiling
    def set_profile(self, enable: bool) -> None:
        native.check(self._lib.ace355_dit_set_profile(self._h, 1 if enable else 0), "dit_set_profile")

    def get_profile(self) -> Dict[str, float]:
        g, gf, a, af, n = C.c_double(), C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        native.check(self._lib.ace355_dit_get_profile(self._h, C.byref(g), C.byref(gf), C.byref(a), C.byref(af), C.byref(n)),
                     "dit_get_profile")
        return {"gemm_ms": g.value, "gemm_flops": gf.value, "attn_ms": a.value, "attn_flops": af.value, "gemm_launches": n.value}


def generate_latents(dit: NativeDit, null_condition_emb: torch.Tensor, encoder_hidden_states: torch.Tensor,
                     context_latents: torch.Tensor, seed=None, infer_method: str = "ode", infer_steps: int = 30,
                     diffusion_guidance_sale: float = 7.0, cfg_interval_start: float = 0.0, cfg_interval_end: float = 1.0,
                     use_adg: bool = False, shift: float = 1.0, timesteps=None, audio_cover_strength: float = 1.0,
                     cover_noise_strength: float = 0.0, src_latents: Optional[torch.Tensor] = None,
                     encoder_hidden_states_non_cover: Optional[torch.Tensor] = None,
                     context_latents_non_cover: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                     sde_noise: Optional[torch.Tensor] = None, sde_next_from_schedule: bool = False) -> Dict:
    """The part of ``generate_audio`` (modeling_acestep_v15_base.py:1861-1989) after ``prepare_condition``.

    The reference replicates ONE caption across the batch (handler/batch_prep.py:93-96), so the cross-attention
    K/V of identical rows of ``encoder_hidden_states`` are computed once per distinct row; batches whose items carry
    different conditions (what the model's own loop accepts, base.py:1905-1911) run as ONE native call with a per-item
    slot table.
    """
    t0 = time.time()
    B, T, _ = context_latents.shape
    enc = encoder_hidden_states
    enc_nc = encoder_hidden_states_non_cover

    def distinct(e):
        """Slot plan for a [B or 1, L, D] condition batch: (rows to upload, per-item index into them).  The reference handler
        replicates ONE caption across the batch (handler/batch_prep.py:93-96): identical rows share one cross-K/V slot."""
        if e.shape[0] == 1 or e.stride(0) == 0:   # one row, or an expand()ed view of one row: nothing to compare, no device sync
            return [0], [0] * B
        if bool((e == e[:1]).all()):
            return [0], [0] * B
        rows, idx = [], []
        for b in range(B):
            for k, r in enumerate(rows):
                if torch.equal(e[b], e[r]):
                    idx.append(k)
                    break
            else:
                rows.append(b)
                idx.append(len(rows) - 1)
        return rows, idx
    ts = schedule(infer_steps, shift, timesteps)
    steps = ts.numel() - 1
    cover_steps = int(steps * audio_cover_strength)
    nearest = None
    if cover_noise_strength > 0.0:  # modeling_acestep_v15_base.py:1879-1900 (the renoised start itself: below, once the noise is drawn)
        eff = 1.0 - cover_noise_strength
        tv = ts[:-1].tolist()
        nearest = min(tv, key=lambda x: abs(x - eff))
        start = tv.index(nearest)
        ts = ts[start:]
        steps = ts.numel() - 1
        cover_steps = int(steps * audio_cover_strength)
    # slot layout of one call: [0, n_c) distinct cover / main conditions, n_c = null, then the distinct non-cover conditions
    rows_c, idx_c = distinct(enc)
    n_c = len(rows_c)
    do_cfg = diffusion_guidance_sale > 1.0
    rows_n, idx_n = [], []
    if cover_steps < steps:
        if enc_nc is None or context_latents_non_cover is None:
            raise ValueError("audio_cover_strength < 1 needs the non-cover conditions")
        # (cover and non-cover conditions may have different encoder lengths, as in the reference, base.py:1916-1927: each phase's
        #  forwards see one length, and the CFG null slot is the same constant whatever length it was expanded to)
        rows_n, idx_n = distinct(enc_nc)
    n_slots = n_c + 1 + len(rows_n)   # (the null slot keeps its place in the layout with CFG off)
    if n_slots > native.MAX_SLOTS:    # checked BEFORE any cross-K/V build, for every path
        raise ValueError(f"too many distinct conditions for one call: {n_c} + null + {len(rows_n)} non-cover > {native.MAX_SLOTS} slots")
    dit.trim_slots(n_slots)           # slots a previous, larger request left resident (150 MB each) are released
    for k, r in enumerate(rows_c):
        dit.set_condition(k, enc[r])
    null_slot = n_c
    if do_cfg:
        dit.set_condition(null_slot, null_condition_emb.reshape(1, -1), L=enc.shape[1])
    ctx_nc, nc_slots = None, None
    if cover_steps < steps:
        for k, r in enumerate(rows_n):
            dit.set_condition(n_c + 1 + k, enc_nc[r])
        nc_slots = [n_c + 1 + k for k in idx_n]
        ctx_nc = context_latents_non_cover
    # the per-song noise (CPU generators, the reference's stream) is drawn AFTER the cross-K/V builds were launched: the host draws while
    # the GPU projects the conditions instead of the GPU waiting for the draw (1.3 ms per 8-song request)
    if noise is None:
        noise = prepare_noise((B, T, context_latents.shape[-1] // 2), seed)
    xt0 = noise if nearest is None else nearest * noise + (1 - nearest) * src_latents.detach().float().cpu()
    t1 = time.time()
    if use_adg and B > 1:
        # the reference's adg_forward only broadcasts for batch 1 (apg_guidance.py:150-168 multiplies [n*t,1] by [n,t,c])
        raise ValueError("use_adg is only defined for batch size 1 in the reference")
    out = dit.sample(xt0, context_latents, ts, diffusion_guidance_sale, cfg_interval_start, cfg_interval_end, infer_method,
                     use_adg, cond_slot=idx_c[0], null_slot=null_slot, cover_switch_step=cover_steps,
                     non_cover_slot=nc_slots[0] if nc_slots else 0, ctx_non_cover=ctx_nc, sde_noise=sde_noise,
                     cond_slots=idx_c, non_cover_slots=nc_slots, sde_next_from_schedule=sde_next_from_schedule)
    dit.poll_errors()   # (synchronises: the reference's timing contract wants the diffusion done here anyway)
    t2 = time.time()
    return {"target_latents": out,
            "time_costs": {"encoder_time_cost": t1 - t0, "diffusion_time_cost": t2 - t1,
                           "diffusion_per_step_time_cost": (t2 - t1) / max(steps, 1), "total_time_cost": t2 - t0}}


# ------------------------------------------------------------------------------------------------ turbo (8-step distilled)
TURBO_VALID_SHIFTS = [1.0, 2.0, 3.0]
TURBO_VALID_TIMESTEPS = [
    1.0, 0.9545454545454546, 0.9333333333333333, 0.9, 0.875, 0.8571428571428571, 0.8333333333333334, 0.7692307692307693, 0.75,
    0.6666666666666666, 0.6428571428571429, 0.625, 0.5454545454545454, 0.5, 0.4, 0.375, 0.3, 0.25, 0.2222222222222222, 0.125]
TURBO_SHIFT_TIMESTEPS = {
    1.0: [1.0, 0.875, 0.75, 0.625, 0.5, 0.375, 0.25, 0.125],
    2.0: [1.0, 0.9333333333333333, 0.8571428571428571, 0.7692307692307693, 0.6666666666666666, 0.5454545454545454, 0.4, 0.2222222222222222],
    3.0: [1.0, 0.9545454545454546, 0.9, 0.8333333333333334, 0.75, 0.6428571428571429, 0.5, 0.3],
}


def turbo_schedule(shift: float = 3.0, timesteps=None) -> List[float]:
    """Timestep table of the turbo model (models/turbo/modeling_acestep_v15_turbo.py:1807-1865): fixed 8-value tables for
    shift in {1,2,3} (other shifts snap to the nearest); explicit ``timesteps`` drop trailing zeros, are cut to 20 entries and
    snapped to the 20 trained values."""
    if timesteps is not None:
        ts = [float(x) for x in (timesteps.tolist() if isinstance(timesteps, torch.Tensor) else list(timesteps))]
        while ts and ts[-1] == 0:
            ts.pop()
        if len(ts) >= 1:
            return [min(TURBO_VALID_TIMESTEPS, key=lambda x: abs(x - t)) for t in ts[:20]]
    return list(TURBO_SHIFT_TIMESTEPS[min(TURBO_VALID_SHIFTS, key=lambda x: abs(x - shift))])


def generate_latents_turbo(dit: NativeDit, encoder_hidden_states: torch.Tensor, context_latents: torch.Tensor, seed=None,
                           shift: float = 3.0, timesteps=None, infer_method: str = "ode", audio_cover_strength: float = 1.0,
                           cover_noise_strength: float = 0.0, src_latents: Optional[torch.Tensor] = None,
                           encoder_hidden_states_non_cover: Optional[torch.Tensor] = None,
                           context_latents_non_cover: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                           sde_noise: Optional[torch.Tensor] = None) -> Dict:
    """The turbo model's ``generate_audio`` after ``prepare_condition`` (turbo.py:1900-1995): same network, no CFG / null
    branch, fixed tables, last step ``x0 = xt - vt * t``: the base loop on ``table + [0]`` with guidance 1."""
    table = turbo_schedule(shift, timesteps)
    return generate_latents(dit, None, encoder_hidden_states, context_latents, seed=seed, infer_method=infer_method,
                            infer_steps=len(table), diffusion_guidance_sale=1.0, shift=shift, timesteps=table + [0.0],
                            audio_cover_strength=audio_cover_strength, cover_noise_strength=cover_noise_strength, src_latents=src_latents,
                            encoder_hidden_states_non_cover=encoder_hidden_states_non_cover,
                            context_latents_non_cover=context_latents_non_cover, noise=noise, sde_noise=sde_noise,
                            sde_next_from_schedule=True)  # "sde": renoise to the next TABLE value (turbo.py:1980-1984)
