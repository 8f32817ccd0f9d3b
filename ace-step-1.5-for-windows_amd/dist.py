"""Data-parallel runner pieces: one process per GPU, songs sharded across ranks, ONE collective per request.

The reference has no multi-GPU inference; this is new functionality shaped by north_star:
"batch-of-songs generation shards data-parallel across the 8 GPUs of one node with RCCL broadcast of text/LM
conditioning over xGMI and per-rank independent samplers".  The conditioning bundle of one request
(encoder states [L,D], null embedding [D], shared context latents [T,128]; a few MB) is produced once on rank 0
and broadcast; every rank then runs its own sampler over its slice of the seed list.  No per-step collective exists.

Backend "nccl" is RCCL on PyTorch-ROCm; CPU tests use "gloo" (tests/test_dist_cpu.py).
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

_MAX_ITEMS = 16
_MAX_DIMS = 4


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


def broadcast_conditioning(bundle: Dict[str, torch.Tensor], src: int = 0) -> Dict[str, torch.Tensor]:
    """Broadcast a dict of fp32 tensors from `src` as ONE flat payload (+ one small int64 header).

    Every rank passes a dict with the same keys in the same order; non-src ranks may pass tensors of any content
    (or wrong shape): shapes travel in the header.  Flat one-hop broadcast is the right algorithm on the xGMI full
    mesh for MB-scale payloads ("Distributed communication backend").
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bundle
    keys = list(bundle.keys())
    assert len(keys) <= _MAX_ITEMS
    rank = dist.get_rank()
    dev = next(iter(bundle.values())).device
    header = torch.zeros(_MAX_ITEMS * (_MAX_DIMS + 1), dtype=torch.int64, device=dev)
    if rank == src:
        h = []
        for k in keys:
            t = bundle[k]
            assert t.dim() <= _MAX_DIMS
            h += [t.dim()] + list(t.shape) + [0] * (_MAX_DIMS - t.dim())
        header[: len(h)] = torch.tensor(h, dtype=torch.int64)
    dist.broadcast(header, src=src)
    hl = header.tolist()
    shapes = []
    for i in range(len(keys)):
        nd = hl[i * (_MAX_DIMS + 1)]
        shapes.append(tuple(hl[i * (_MAX_DIMS + 1) + 1: i * (_MAX_DIMS + 1) + 1 + nd]))
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    if rank == src:
        torch.cat([bundle[k].detach().reshape(-1).to(torch.float32) for k in keys], out=flat)
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for k, s, n in zip(keys, shapes, sizes):
        out[k] = flat[off: off + n].view(s)
        off += n
    return out


def gather_waveforms(wav: torch.Tensor, dst: int = 0):
    """Optional final gather of per-rank waveforms [b_r, 2, samples] to `dst` (list of tensors there, None elsewhere)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [wav]
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [torch.zeros(1, dtype=torch.int64, device=wav.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([wav.shape[0]], dtype=torch.int64, device=wav.device))
    out = None
    if rank == dst:
        out = [torch.empty((int(c.item()),) + tuple(wav.shape[1:]), dtype=wav.dtype, device=wav.device) for c in counts]
    if dist.get_backend() == "nccl":
        # RCCL has no gather primitive in torch for uneven sizes: point-to-point sends into dst
        if rank == dst:
            out[dst].copy_(wav)
            for r in range(world):
                if r != dst and out[r].numel():
                    dist.recv(out[r], src=r)
        elif wav.numel():
            dist.send(wav.contiguous(), dst=dst)
    else:
        if rank == dst:
            out[dst].copy_(wav)
            for r in range(world):
                if r != dst and out[r].numel():
                    dist.recv(out[r], src=r)
        elif wav.numel():
            dist.send(wav.contiguous(), dst=dst)
    return out
