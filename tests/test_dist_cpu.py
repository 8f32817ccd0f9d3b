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
    mine = ref if rank == 0 else {"enc": None, "null": None, "ctx": None}  # only rank 0 knows the request: shapes travel in the payload
    out = a_dist.broadcast_conditioning(mine, src=0, capacity_bytes=1 << 16)
    ok = all(torch.equal(out[k], ref[k]) for k in ref)
    out2 = a_dist.broadcast_conditioning(mine, src=0, capacity_bytes=1 << 16)  # the persistent buffer is reused: results are copies
    ok = ok and all(torch.equal(out[k], ref[k]) and torch.equal(out2[k], ref[k]) for k in ref)
    try:  # a bundle that does not fit raises on EVERY rank (no dead-lock)
        a_dist.broadcast_conditioning({"big": torch.zeros(1 << 15) if rank == 0 else None}, src=0, capacity_bytes=1 << 16)
        ok = False
    except ValueError:
        pass
    # per-item LM hints [G, T, 64] scattered by song ownership (G = 11 over 2 ranks: 6 + 5 rows)
    G, T = 11, 9
    hints = torch.arange(G * T * 64, dtype=torch.float32).view(G, T, 64) if rank == 0 else None
    mine_h = a_dist.scatter_lm_hints(hints, G, T, 64, src=0, device=torch.device("cpu"))
    s0, e0 = a_dist.shard_range(G, world, rank)
    ok = ok and torch.equal(mine_h, torch.arange(G * T * 64, dtype=torch.float32).view(G, T, 64)[s0:e0])
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
    h = torch.zeros(3, 4, 64)
    assert a_dist.scatter_lm_hints(h, 3, 4) is h
    assert a_dist.gather_waveforms(torch.zeros(1, 2, 3))[0].shape == (1, 2, 3)


def test_bench_gpus_flag_spawns_ranks_dry_run():
    """`python bench.py --gpus 2` started as a plain process must launch 2 ranks itself (round 1 parsed --gpus and ignored it).
    --dry-run keeps the launcher, the sharding and the collective sequence (one broadcast + one scatter per pass, barrier,
    MAX-reduce) and swaps the GPU work for a memcpy, over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    for scaling, want_g, want_b0 in (("weak", 16, 8), ("strong", 8, 4)):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run",
                            "--lm-hints", "--scaling", scaling, "--duration", "2"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout  # rank 0 only
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["dry_run"] and out["collectives_ok"] and out["scaling"] == scaling
        assert out["config"]["global_batch"] == want_g and out["config"]["batch_rank0"] == want_b0
