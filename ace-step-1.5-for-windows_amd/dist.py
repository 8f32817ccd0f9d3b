"""Data-parallel runner: one process per GPU, songs sharded across ranks, the conditioning bundle broadcast once per request.

The reference has no multi-GPU inference; this is new functionality shaped by north_star:
"batch-of-songs generation shards data-parallel across the 8 GPUs of one node with RCCL broadcast of text/LM
conditioning over xGMI and per-rank independent samplers".  The conditioning bundle of one request (distinct encoder-state
rows [n, L, D], null embedding [D], context latents [1 | G, T, 128], per-item seeds, request knobs) is produced once on rank 0
and broadcast at its exact size; every rank then runs the single-GPU path (sample -> decode -> normalise) over its contiguous
slice of the song list.  Per-item LM hints (``precomputed_lm_hints_25Hz [G,T,64]``, modeling_acestep_v15_base.py:1638-1649) are
scattered instead: each rank receives only its rows.  No per-step collective exists.

``run_request`` is the composition (broadcast -> shard -> execute -> gather); ``bench.py --gpus N``,
``NativeHandler.generate_music(..., data_parallel=True)`` and tests/test_dist_cpu.py (gloo, world 2) all drive this one
function with their own ``execute`` callable.

Backend "nccl" is RCCL on PyTorch-ROCm; CPU tests use "gloo".
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_MAX_ITEMS = 16
_MAX_DIMS = 4
# header (int32 words): [magic, n_items | -1, payload bytes, error code, (dtype code, ndim, d0..d3) x items, host area]
# An item whose dtype code is >= _IN_HEADER travels INSIDE the header's host area (its words in item order) instead of in the
# payload: the small per-request scalars (seeds, caption index, knobs, explicit timesteps) are read on the host by every rank anyway,
# and the header is the one tensor a receiver reads back - so a request costs ONE device -> host read, not one per scalar tensor
# (round 3 read three small tensors back with .tolist(): three more stream syncs per request at N > 1).
_ITEMS_END = 4 + _MAX_ITEMS * (_MAX_DIMS + 2)
_HOST_WORDS = 1024
_HEADER = _ITEMS_END + _HOST_WORDS
_IN_HEADER = 100
_MAGIC = 0x0ACE0356
_ALIGN = 16
# Sanity bound a receiver accepts for one bundle, not a feature limit: a 600 s request with L = 2305 is 13 MB in bf16, 64 songs x 240 s
# with per-song contexts (LM hints, cover sources) 98 MB (advisor r3: the round-3 value of 64 MB refused that request on every rank).
DEFAULT_CAPACITY_BYTES = 1 << 30
_DTYPES = [torch.float32, torch.bfloat16, torch.int64, torch.int32, torch.float64]
_ERR_NONE, _ERR_TOO_BIG, _ERR_BAD_ITEM, _ERR_NO_REQUEST = 0, 1, 2, 3
# keys whose (small) tensors ride in the header and come back as CPU tensors on every rank
HOST_KEYS = ("seeds", "enc_index", "knobs", "timesteps", "enc_index_non_cover")

# What travels in bf16: every tensor the native path rounds to bf16 on arrival anyway (encoder states and the null embedding in
# ace355_dit_set_condition, context latents in pack_xin / set_xin_ctx; csrc/dit.hip) - the transport rounding (RNE, same as the
# kernels') is the one rounding, so results are bit-identical to a single-GPU run of the same songs.
BF16_KEYS = ("enc", "enc_rows", "null", "ctx", "ctx_non_cover", "enc_rows_non_cover")


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun-style env; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend)
    return rank, world, local_rank


def shard_range(global_batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of the global song list owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_seeds(seeds: List[int], world: int, rank: int) -> List[int]:
    """Per-rank slice of the request's seed list (handler/task_utils.py:19-57 produces the global list)."""
    s, e = shard_range(len(seeds), world, rank)
    return list(seeds[s:e])


def host_staged() -> bool:
    """True under the gloo backend: torch's gloo moves CUDA tensors for broadcast / all_reduce only (send, recv, scatter and
    all_gather are CPU-only there), so every collective buffer of this module then lives in host memory and results are moved to
    the GPU by their consumer.  That is the configuration of the CPU tests and of the two-ranks-on-one-GPU test
    (tests/test_dist_gpu.py); under nccl (= RCCL) the buffers are device tensors and nothing is staged."""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def _collective_device(device: Optional[torch.device], bundle: Dict[str, Any]) -> torch.device:
    """Where the collective buffers live.  Under nccl / RCCL they must be GPU tensors on EVERY rank: a receiver that was handed
    only None values (and no `device`) takes the current CUDA device, never the CPU (src would block forever on a mixed call).
    Under gloo: always the host (`host_staged`)."""
    if host_staged():
        return torch.device("cpu")
    if device is not None:
        return torch.device(device)
    for t in bundle.values():
        if torch.is_tensor(t):
            if dist.get_backend() != "nccl" or t.is_cuda:
                return t.device
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _pad(n: int) -> int:
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


def _to_words(t: torch.Tensor) -> List[int]:
    """A small CPU tensor of one of _DTYPES (not bf16) as int32 words."""
    import numpy as np
    a = t.detach().cpu().contiguous().numpy()
    return np.frombuffer(a.tobytes(), dtype=np.int32).tolist()


def _from_words(words: List[int], dt: torch.dtype, shape) -> torch.Tensor:
    import numpy as np
    a = np.array(words, dtype=np.int32).tobytes()
    npdt = {torch.float32: np.float32, torch.int64: np.int64, torch.int32: np.int32, torch.float64: np.float64}[dt]
    return torch.from_numpy(np.frombuffer(a, dtype=npdt).copy()).reshape(shape)


def broadcast_conditioning(bundle: Dict[str, Optional[torch.Tensor]], src: int = 0, capacity_bytes: int = DEFAULT_CAPACITY_BYTES,
                           device: Optional[torch.device] = None, bf16_keys: Sequence[str] = BF16_KEYS,
                           host_keys: Sequence[str] = HOST_KEYS, src_error: int = 0) -> Dict[str, torch.Tensor]:
    """Broadcast a dict of tensors from `src` at its EXACT size: a 0.4 KB header (shapes, dtypes, byte count - only `src` knows
    the request: L depends on the caption) followed by one flat byte payload, 16-byte aligned items.  Keys in `bf16_keys` travel
    as bf16 (see BF16_KEYS: lossless for this path), integer tensors as they are, the rest as fp32.  A 30 s request is 3.3 MB
    (enc 769 x 2048 + ctx 750 x 128 in bf16) against the 32 MB fp32 buffer of round 2; flat one-hop broadcast is the right
    algorithm on the xGMI full mesh for MB-scale payloads.

    Non-src ranks pass the KEYS (values ignored, may be None).  Anything wrong with the bundle on `src` (too many items or
    dimensions, larger than `capacity_bytes`, or `src_error`: the caller on `src` has no request to send) travels in the header and
    raises on EVERY rank instead of dead-locking the others.  Keys in `host_keys` whose tensors fit the header's host area (4 KB in
    all; int / fp32 / fp64) travel inside the header and come back as CPU tensors on every rank - no extra device read.
    Returned tensors keep their transport dtype (bf16 / int / fp32) and are private copies."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bundle
    keys = list(bundle.keys())
    rank = dist.get_rank()
    dev = _collective_device(device, bundle)
    head = torch.zeros(_HEADER, dtype=torch.int32, device=dev)
    parts: List[Optional[torch.Tensor]] = []
    total = 0
    if rank == src:
        h = [_MAGIC, len(keys), 0, _ERR_NONE]
        host: List[int] = []
        err = src_error
        if err == _ERR_NONE and len(keys) > _MAX_ITEMS:
            err = _ERR_BAD_ITEM
        elif err == _ERR_NONE:
            for k in keys:
                t = bundle[k]
                if not torch.is_tensor(t) or t.dim() > _MAX_DIMS:
                    err = _ERR_BAD_ITEM
                    break
                t = t.detach()
                if k in bf16_keys and t.is_floating_point():
                    t = t.to(torch.bfloat16)
                elif t.dtype not in _DTYPES:
                    t = t.to(torch.float32) if t.is_floating_point() else t.to(torch.int64)
                nwords = (t.numel() * t.element_size() + 3) // 4
                if k in host_keys and t.dtype != torch.bfloat16 and (t.numel() * t.element_size()) % 4 == 0 and len(host) + nwords <= _HOST_WORDS:
                    h += [_IN_HEADER + _DTYPES.index(t.dtype), t.dim()] + list(t.shape) + [0] * (_MAX_DIMS - t.dim())
                    host += _to_words(t)
                    parts.append(None)
                    continue
                t = t.to(dev).contiguous()
                h += [_DTYPES.index(t.dtype), t.dim()] + list(t.shape) + [0] * (_MAX_DIMS - t.dim())
                parts.append(t)
                total += _pad(t.numel() * t.element_size())
            if err == _ERR_NONE and total > capacity_bytes:
                err = _ERR_TOO_BIG
        if err != _ERR_NONE:
            h = [_MAGIC, -1, 0, err]
            parts, total, host = [], 0, []
        else:
            h[2] = total
        head[: len(h)] = torch.tensor(h, dtype=torch.int32)
        if host:
            head[_ITEMS_END: _ITEMS_END + len(host)] = torch.tensor(host, dtype=torch.int32)
    dist.broadcast(head, src=src)
    hl = head.tolist()   # (the one host read of a request: receivers cannot slice the payload without the shapes)
    if hl[0] != _MAGIC:
        raise RuntimeError("broadcast_conditioning: ranks disagree on the header layout")
    if hl[1] < 0:
        if hl[3] == _ERR_TOO_BIG:
            raise ValueError(f"broadcast_conditioning: the bundle does not fit the {capacity_bytes}-byte limit")
        if hl[3] == _ERR_NO_REQUEST:
            raise ValueError("broadcast_conditioning: the source rank holds no request")
        raise ValueError(f"broadcast_conditioning: at most {_MAX_ITEMS} tensors of at most {_MAX_DIMS} dimensions per bundle")
    if hl[1] != len(keys):
        raise RuntimeError("broadcast_conditioning: ranks passed different key lists")
    total = hl[2]
    if total > capacity_bytes:
        raise ValueError(f"broadcast_conditioning: the bundle does not fit the {capacity_bytes}-byte limit")
    buf = torch.empty(total, dtype=torch.uint8, device=dev)
    if rank == src:
        off = 0
        for t in parts:
            if t is None:
                continue
            nb = t.numel() * t.element_size()
            buf[off: off + nb] = t.reshape(-1).view(torch.uint8)
            off += _pad(nb)
    if total:
        dist.broadcast(buf, src=src)
    out, off, hoff = {}, 0, _ITEMS_END
    for i, k in enumerate(keys):
        base = 4 + i * (_MAX_DIMS + 2)
        code, nd = hl[base], hl[base + 1]
        shape = tuple(hl[base + 2: base + 2 + nd])
        if code >= _IN_HEADER:   # rode in the header: a CPU tensor on every rank (src included: same object kind everywhere)
            dt = _DTYPES[code - _IN_HEADER]
            nw = int(torch.Size(shape).numel()) * torch.empty((), dtype=dt).element_size() // 4
            out[k] = _from_words(hl[hoff: hoff + nw], dt, shape)
            hoff += nw
            continue
        dt = _DTYPES[code]
        nb = int(torch.Size(shape).numel()) * torch.empty((), dtype=dt).element_size()
        out[k] = buf[off: off + nb].view(dt).view(shape).clone() if rank != src else parts[i]
        off += _pad(nb)
    return out


def scatter_lm_hints(hints: Optional[torch.Tensor], global_batch: int, T: int, channels: int = 64, src: int = 0,
                     device: Optional[torch.device] = None) -> torch.Tensor:
    """Per-item LM hints ``precomputed_lm_hints_25Hz [G, T, 64]`` (think-mode planner output, one row per song;
    modeling_acestep_v15_base.py:1638-1649) live on `src`; every rank gets the rows of ITS songs (`shard_range`), in one
    `scatter` collective.  G, T travel in the conditioning bundle, so every rank can size its receive buffer.  Ranks own
    at most ceil(G / world) rows; shorter slices are padded in flight and trimmed on arrival."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        assert hints is not None
        return hints
    world, rank = dist.get_world_size(), dist.get_rank()
    rows_max = -(-global_batch // world)
    device = _collective_device(device, {"hints": hints})
    recv = torch.empty(rows_max, T, channels, dtype=torch.float32, device=device)
    parts = None
    if rank == src:
        assert hints is not None and tuple(hints.shape) == (global_batch, T, channels)
        hints = hints.detach().to(device=device, dtype=torch.float32)
        parts = []
        for r in range(world):
            s, e = shard_range(global_batch, world, r)
            p = torch.zeros(rows_max, T, channels, dtype=torch.float32, device=device)
            p[: e - s] = hints[s:e]
            parts.append(p)
    dist.scatter(recv, parts, src=src)
    s, e = shard_range(global_batch, world, rank)
    return recv[: e - s]


def gather_waveforms(wav: torch.Tensor, dst: int = 0):
    """Optional final gather of per-rank waveforms [b_r, 2, samples] to `dst` (list of tensors there, None elsewhere).
    Slices may be uneven, which torch's `gather` cannot express on any backend: sizes go through one small all_gather, the
    payloads point-to-point into `dst` (one hop each on the xGMI mesh)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [wav]
    world, rank = dist.get_world_size(), dist.get_rank()
    if host_staged():   # (gloo: point-to-point and all_gather are CPU-only; the gathered list then lives in host memory)
        wav = wav.cpu()
    # [songs, samples per song]: `dst` sizes its receive buffers from the SENDER's numbers (a rank without a song has no duration)
    n = torch.tensor([wav.shape[0], wav.shape[-1] if wav.dim() > 1 else 1], dtype=torch.int64, device=wav.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    if rank != dst:
        if wav.numel():
            dist.send(wav.contiguous(), dst=dst)
        return None
    out = []
    for r, c in enumerate(counts):
        b, smp = int(c[0]), int(c[1])
        shape = (b,) + (tuple(wav.shape[1:-1]) + (smp,) if wav.dim() > 1 else ())
        out.append(torch.empty(shape, dtype=wav.dtype, device=wav.device))
    out[dst].copy_(wav)
    for r in range(world):
        if r != dst and out[r].numel():
            dist.recv(out[r], src=r)
    return out


# ------------------------------------------------------------------------------------------------ the request runner
# every per-request scalar of generate_music travels with the request: a rank called with other defaults must not generate its songs
# with settings that differ from rank 0's (advisor r3); explicit `timesteps` ride as their own (header) item
KNOBS = ("inference_steps", "guidance_scale", "shift", "cfg_interval_start", "cfg_interval_end", "use_adg", "infer_method_sde",
         "use_tiled_decode", "latent_shift", "latent_rescale", "audio_cover_strength", "cover_noise_strength")
_KNOB_DEFAULTS = {"inference_steps": 27, "guidance_scale": 7.0, "shift": 1.0, "cfg_interval_start": 0.0, "cfg_interval_end": 1.0,
                  "use_adg": 0.0, "infer_method_sde": 0.0, "use_tiled_decode": 1.0, "latent_shift": 0.0, "latent_rescale": 1.0,
                  "audio_cover_strength": 1.0, "cover_noise_strength": 0.0}


def _distinct_rows(enc: torch.Tensor, G: int) -> Tuple[List[int], List[int]]:
    """(rows, index): the distinct rows of a [G | 1, ...] tensor and, per song, which of them it uses."""
    rows: List[int] = []
    idx: List[int] = []
    for b in range(G):
        r0 = b if enc.shape[0] > 1 else 0
        for k, r in enumerate(rows):
            if r == r0 or torch.equal(enc[r0], enc[r]):
                idx.append(k)
                break
        else:
            rows.append(r0)
            idx.append(len(rows) - 1)
    return rows, idx


def pack_request(encoder_hidden_states: torch.Tensor, context_latents: torch.Tensor, seeds: Sequence[int],
                 null_condition_emb: Optional[torch.Tensor] = None, timesteps: Optional[Sequence[float]] = None,
                 src_latents: Optional[torch.Tensor] = None, encoder_hidden_states_non_cover: Optional[torch.Tensor] = None,
                 context_latents_non_cover: Optional[torch.Tensor] = None, **knobs) -> Dict[str, torch.Tensor]:
    """The wire form of one generate_music request of G songs (rank 0): the DISTINCT rows of ``encoder_hidden_states [G | 1, L, D]``
    with an index per song (the reference replicates one caption across the batch, handler/batch_prep.py:93-96: one row),
    ``context_latents [G | 1, T, 128]`` collapsed to one row when every song shares it, per-song seeds, the request knobs.
    A cover / repaint request adds its tensors (round 5, advisor r4): ``src_latents [G | 1, T, 64]`` (fp32: the renoised start
    ``t x noise + (1 - t) x src`` of base.py:1879-1900 is host arithmetic in fp32), the non-cover conditions
    ``encoder_hidden_states_non_cover [G | 1, L', D]`` (distinct rows + index, like the cover ones) and
    ``context_latents_non_cover [G | 1, T, 128]``; absent ones travel as empty tensors, so every rank passes one key list."""
    G = len(seeds)
    enc = encoder_hidden_states if encoder_hidden_states.dim() == 3 else encoder_hidden_states[None]
    rows, idx = _distinct_rows(enc, G)
    ctx = context_latents if context_latents.dim() == 3 else context_latents[None]
    if ctx.shape[0] > 1 and bool((ctx == ctx[:1]).all()):
        ctx = ctx[:1]
    unknown = set(knobs) - set(KNOBS)
    if unknown:
        raise ValueError(f"pack_request: unknown knobs {sorted(unknown)}")
    kv = dict(_KNOB_DEFAULTS)
    kv.update({k: float(v) for k, v in knobs.items()})
    b = {"enc_rows": enc[rows].contiguous(), "enc_index": torch.tensor(idx, dtype=torch.int32),
         "ctx": ctx.contiguous(), "seeds": torch.tensor([int(s) for s in seeds], dtype=torch.int64),
         "knobs": torch.tensor([kv[k] for k in KNOBS], dtype=torch.float64)}
    if null_condition_emb is not None:
        b["null"] = null_condition_emb.reshape(-1).contiguous()
    # explicit schedule of the sft variant (base.py:1864-1875); empty = derive it from inference_steps / shift
    b["timesteps"] = torch.as_tensor([] if timesteps is None else [float(t) for t in timesteps], dtype=torch.float32)
    for name, t, ch in (("src", src_latents, ctx.shape[-1] // 2), ("ctx_non_cover", context_latents_non_cover, ctx.shape[-1])):
        if t is None:
            b[name] = torch.zeros(0)
            continue
        t = t if t.dim() == 3 else t[None]
        if t.shape[0] not in (1, G):
            raise ValueError(f"pack_request: {name} must have 1 or {G} rows, got {t.shape[0]}")
        if t.shape[1] != ctx.shape[1] or t.shape[2] != ch:   # (a mismatch would otherwise ship and fail later on EVERY rank: advisor r5)
            raise ValueError(f"pack_request: {name} must be [*, {ctx.shape[1]}, {ch}] like the request's context latents, got {tuple(t.shape)}")
        if t.shape[0] > 1 and bool((t == t[:1]).all()):
            t = t[:1]
        b[name] = t.contiguous()
    if encoder_hidden_states_non_cover is None:
        b["enc_rows_non_cover"], b["enc_index_non_cover"] = torch.zeros(0), torch.zeros(0, dtype=torch.int32)
    else:
        e2 = encoder_hidden_states_non_cover if encoder_hidden_states_non_cover.dim() == 3 else encoder_hidden_states_non_cover[None]
        if e2.shape[0] not in (1, G):
            raise ValueError(f"pack_request: encoder_hidden_states_non_cover must have 1 or {G} rows, got {e2.shape[0]}")
        r2, i2 = _distinct_rows(e2, G)
        b["enc_rows_non_cover"], b["enc_index_non_cover"] = e2[r2].contiguous(), torch.tensor(i2, dtype=torch.int32)
    return b


_REQUEST_KEYS = ("enc_rows", "enc_index", "ctx", "seeds", "knobs", "null", "timesteps",
                 "src", "ctx_non_cover", "enc_rows_non_cover", "enc_index_non_cover")
_OPTIONAL_KEYS = ("null", "timesteps", "src", "ctx_non_cover", "enc_rows_non_cover", "enc_index_non_cover")


def run_request(request: Optional[Dict[str, torch.Tensor]], execute: Callable[[Dict[str, Any]], Any], src: int = 0,
                device: Optional[torch.device] = None, lm_hints: Optional[torch.Tensor] = None, use_lm_hints: bool = False,
                gather: bool = False, capacity_bytes: int = DEFAULT_CAPACITY_BYTES) -> Dict[str, Any]:
    """One generate_music request over all ranks: ONE exact-size broadcast of the packed request (`pack_request`, known on `src`
    only; the others pass None), contiguous song slices (`shard_range`: 8/4/2/1 songs per rank for the metric's batch of 8 on
    1/2/4/8 GPUs), `execute(local)` on every rank that owns a song, optional gather of what it returned.

    `local` = {"encoder_hidden_states" [b, L, D], "context_latents" [b, T, 128], "null_condition_emb" [D] | None, "seeds" [b],
    "knobs" {name: float}, "timesteps" list | None, "range" (s0, s1), "global_batch" G, "enc_rows", "enc_index"} - tensors in their
    transport dtype on the collective device.  A source rank without a request makes EVERY rank raise (in-band, like the other refusals).  With `use_lm_hints` (same value on every rank) the per-song hints [G, T, 64] on `src` are scattered and
    replace the source-latent half of each song's context (base.py:1646-1649).  Returns {"local": execute's result or None,
    "range", "global_batch", "gathered": list per rank on `src` when `gather` (execute must then return a [b, ...] tensor)}."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world > 1:
        keys = list(_REQUEST_KEYS)
        src_error = _ERR_NONE
        if rank == src and request is not None:
            # (a request without a null embedding / explicit timesteps still ships the keys: every rank must pass the same key list)
            req = {k: request.get(k) for k in keys}
            for k in _OPTIONAL_KEYS:
                if req[k] is None:
                    req[k] = torch.zeros(0, dtype=torch.int32) if k.startswith("enc_index") else torch.zeros(0)
        else:
            # (src without a request: raising HERE would leave the other ranks waiting in the broadcast - the refusal rides in the
            #  header and raises on every rank, advisor r3)
            req = {k: None for k in keys}
            if rank == src:
                src_error = _ERR_NO_REQUEST
        b = broadcast_conditioning(req, src=src, capacity_bytes=capacity_bytes, device=device, src_error=src_error)
    else:
        if request is None:
            raise ValueError("run_request: no request")
        b = dict(request)
        for k in _OPTIONAL_KEYS:
            b.setdefault(k, None)
    G = int(b["seeds"].numel())
    s0, s1 = shard_range(G, world, rank)
    # (seeds / enc_index / knobs / timesteps rode in the header at N > 1: CPU tensors, no device read here)
    kvals = [float(x) for x in b["knobs"].tolist()]
    knobs = dict(_KNOB_DEFAULTS)
    knobs.update(dict(zip(KNOBS, kvals)))   # (a shorter knob vector = an older packer: the missing ones keep their defaults)
    ts_t = b.get("timesteps")
    timesteps = [float(x) for x in ts_t.tolist()] if ts_t is not None and ts_t.numel() else None
    seeds_all = [int(x) for x in b["seeds"].tolist()]
    idx_all = [int(x) for x in b["enc_index"].tolist()]
    ctx = b["ctx"]
    T = ctx.shape[1]
    mine_h = None
    if use_lm_hints:
        if world > 1:
            mine_h = scatter_lm_hints(lm_hints, G, T, ctx.shape[-1] // 2, src=src, device=ctx.device)
        else:
            mine_h = lm_hints[s0:s1]
    result = None
    if s1 > s0:
        ctx_l = (ctx[s0:s1] if ctx.shape[0] > 1 else ctx.expand(s1 - s0, -1, -1)).contiguous()
        if mine_h is not None:
            ctx_l = torch.cat([mine_h.to(ctx_l.dtype), ctx_l[..., ctx_l.shape[-1] // 2:]], -1).contiguous()
        null = b.get("null")
        def rows_of(t):   # this rank's songs of an optional [G | 1, ...] item (None when the request has none)
            if t is None or t.numel() == 0:
                return None
            return (t[s0:s1] if t.shape[0] > 1 else t.expand(s1 - s0, *t.shape[1:])).contiguous()

        enc_nc = None
        rows_nc, idx_nc = b.get("enc_rows_non_cover"), b.get("enc_index_non_cover")
        if rows_nc is not None and rows_nc.numel():
            idx_nc_all = [int(x) for x in idx_nc.tolist()]
            enc_nc = rows_nc[[idx_nc_all[i] for i in range(s0, s1)]]
        local = {"encoder_hidden_states": b["enc_rows"][[idx_all[i] for i in range(s0, s1)]], "context_latents": ctx_l,
                 "null_condition_emb": None if null is None or null.numel() == 0 else null, "seeds": seeds_all[s0:s1], "knobs": knobs,
                 "timesteps": timesteps,
                 "range": (s0, s1), "global_batch": G, "enc_rows": b["enc_rows"], "enc_index": idx_all[s0:s1],
                 "src_latents": rows_of(b.get("src")), "context_latents_non_cover": rows_of(b.get("ctx_non_cover")),
                 "encoder_hidden_states_non_cover": enc_nc}
        result = execute(local)
    out = {"local": result, "range": (s0, s1), "global_batch": G}
    if gather:
        if world > 1:
            if result is None:   # this rank owns no song (G < world): it still takes part in the size exchange
                cdev = torch.device("cpu") if host_staged() else ctx.device
                counts = [torch.zeros(2, dtype=torch.int64, device=cdev) for _ in range(world)]
                dist.all_gather(counts, torch.zeros(2, dtype=torch.int64, device=cdev))
                out["gathered"] = None
            else:
                out["gathered"] = gather_waveforms(result, dst=src)
        else:
            out["gathered"] = [result]
    return out
