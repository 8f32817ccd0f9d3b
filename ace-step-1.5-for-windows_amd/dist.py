"""Data-parallel runner pieces: one process per GPU, songs sharded across ranks, ONE collective per request.

The reference has no multi-GPU inference; this is new functionality shaped by north_star:
"batch-of-songs generation shards data-parallel across the 8 GPUs of one node with RCCL broadcast of text/LM
conditioning over xGMI and per-rank independent samplers".  The conditioning bundle of one request
(encoder states [L,D], null embedding [D], shared context latents [T,128], request scalars; a few MB) is produced once on
rank 0 and broadcast; every rank then runs its own sampler over its slice of the seed list.  Per-item LM hints
(``precomputed_lm_hints_25Hz [G,T,64]``, modeling_acestep_v15_base.py:1638-1649) are scattered instead: each rank receives
only its rows.  No per-step collective exists.

Backend "nccl" is RCCL on PyTorch-ROCm; CPU tests use "gloo" (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

_MAX_ITEMS = 16
_MAX_DIMS = 4
_HEADER = 2 + _MAX_ITEMS * (_MAX_DIMS + 1)      # int32 words: [magic, n_items, (ndim, d0..d3) x items]
_MAGIC = 0x0ACE0355
DEFAULT_CAPACITY_BYTES = 32 << 20               # covers L = 2305 encoder rows (18.9 MB) + a 600 s context (7.7 MB)


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


_bcast_buf: Dict[Tuple[str, int], torch.Tensor] = {}


def broadcast_conditioning(bundle: Dict[str, torch.Tensor], src: int = 0, capacity_bytes: int = DEFAULT_CAPACITY_BYTES,
                           device: Optional[torch.device] = None) -> Dict[str, torch.Tensor]:
    """Broadcast a dict of fp32 tensors from `src` in ONE collective.

    Only `src` knows the request (shapes included: L depends on the caption), so the other ranks cannot size a receive
    buffer from a separate header without a second collective.  Instead every rank keeps one persistent buffer of
    `capacity_bytes` (same value on every rank); `src` writes [header | payload] into it and the WHOLE buffer is
    broadcast: shapes ride in the first words (int32 bit patterns next to the fp32 payload, no value conversion), the
    unused tail costs ~0.2 ms of xGMI time per 32 MB against ~500 ms of compute per request.  Flat one-hop broadcast is the
    right algorithm on the xGMI full mesh for MB-scale payloads.  Non-src ranks pass the KEYS (values
    ignored, may be None); a bundle that does not fit raises on every rank (the error travels in the header) instead of
    dead-locking the others.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bundle
    keys = list(bundle.keys())
    assert len(keys) <= _MAX_ITEMS
    rank = dist.get_rank()
    if device is None:
        device = next((t.device for t in bundle.values() if torch.is_tensor(t)), torch.device("cpu"))
    cap_words = capacity_bytes // 4
    buf = _bcast_buf.get((str(device), cap_words))
    if buf is None:
        buf = torch.zeros(cap_words, dtype=torch.int32, device=device)
        _bcast_buf[(str(device), cap_words)] = buf
    if rank == src:
        h = [_MAGIC, len(keys)]
        total = 0
        for k in keys:
            t = bundle[k]
            assert t.dim() <= _MAX_DIMS
            h += [t.dim()] + list(t.shape) + [0] * (_MAX_DIMS - t.dim())
            total += t.numel()
        if _HEADER + total > cap_words:
            h[1] = -1  # does not fit: tell every rank
        else:
            flat = torch.cat([bundle[k].detach().reshape(-1).to(device=device, dtype=torch.float32) for k in keys])
            buf[_HEADER: _HEADER + total] = flat.view(torch.int32)
        buf[: len(h)] = torch.tensor(h, dtype=torch.int32, device=device)
    dist.broadcast(buf, src=src)
    hl = buf[:_HEADER].tolist()
    if hl[0] != _MAGIC:
        raise RuntimeError("broadcast_conditioning: ranks disagree on the buffer layout (capacity_bytes must match)")
    if hl[1] < 0:
        raise ValueError(f"broadcast_conditioning: the bundle does not fit the {capacity_bytes}-byte broadcast buffer")
    if hl[1] != len(keys):
        raise RuntimeError("broadcast_conditioning: ranks passed different key lists")
    out, off = {}, _HEADER
    for i, k in enumerate(keys):
        nd = hl[2 + i * (_MAX_DIMS + 1)]
        shape = tuple(hl[3 + i * (_MAX_DIMS + 1): 3 + i * (_MAX_DIMS + 1) + nd])
        n = int(torch.Size(shape).numel())
        out[k] = buf[off: off + n].view(torch.float32).view(shape).clone()  # the buffer is reused by the next request
        off += n
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
    if device is None:
        device = hints.device if hints is not None else torch.device("cpu")
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
    counts = [torch.zeros(1, dtype=torch.int64, device=wav.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([wav.shape[0]], dtype=torch.int64, device=wav.device))
    if rank != dst:
        if wav.numel():
            dist.send(wav.contiguous(), dst=dst)
        return None
    out = [torch.empty((int(c.item()),) + tuple(wav.shape[1:]), dtype=wav.dtype, device=wav.device) for c in counts]
    out[dst].copy_(wav)
    for r in range(world):
        if r != dst and out[r].numel():
            dist.recv(out[r], src=r)
    return out
