"""CPU: the data-parallel pieces over gloo with world_size 2 (the N>1 path of bench.py / ace355.dist)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import ace355  # noqa: F401
    from ace355 import dist as a_dist
    r, w, lr = a_dist.init_from_env("gloo")
    assert (r, w, lr) == (rank, world, rank)
    g = torch.Generator().manual_seed(123)
    ref = {"enc": torch.randn(7, 16, generator=g), "null": torch.randn(16, generator=g), "ctx": torch.randn(5, 128, generator=g)}
    mine = ref if rank == 0 else {"enc": torch.empty(1), "null": torch.empty(2, 2), "ctx": torch.zeros(3)}  # shapes travel in the header
    out = a_dist.broadcast_conditioning(mine, src=0)
    ok = all(torch.equal(out[k], ref[k]) for k in ref)
    seeds = list(range(1000, 1011))
    mine_seeds = a_dist.shard_seeds(seeds, world, rank)
    wav = torch.full((len(mine_seeds), 2, 6), float(rank))
    gathered = a_dist.gather_waveforms(wav, dst=0)
    if rank == 0:
        ok = ok and [t.shape[0] for t in gathered] == [6, 5] and float(gathered[1].mean()) == 1.0
    else:
        ok = ok and gathered is None
    dist.barrier()
    q.put((rank, ok, mine_seeds))
    dist.destroy_process_group()


def test_broadcast_shard_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(ok for _r, ok, _s in res)
    assert res[0][2] + res[1][2] == list(range(1000, 1011))  # contiguous, complete, disjoint


def test_shard_range_properties():
    from ace355.dist import shard_range
    for G in (1, 7, 8, 32, 64):
        for W in (1, 2, 4, 8):
            spans = [shard_range(G, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_passthrough():
    from ace355 import dist as a_dist
    b = {"x": torch.ones(2)}
    assert a_dist.broadcast_conditioning(b) is b
    assert a_dist.gather_waveforms(torch.zeros(1, 2, 3))[0].shape == (1, 2, 3)
