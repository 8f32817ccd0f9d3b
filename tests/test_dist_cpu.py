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
    ref = {"enc": torch.randn(7, 16, generator=g), "null": torch.randn(16, generator=g), "ctx": torch.randn(5, 128, generator=g),
           "aux": torch.randn(3, 2, 2, generator=g), "seeds": torch.arange(5, dtype=torch.int64) + (1 << 40), "knobs": torch.randn(4, generator=g).double()}
    # what arrives: the bf16 keys in bf16 (the native path rounds them to bf16 anyway), everything else bit for bit
    want = {k: (v.to(torch.bfloat16) if k in a_dist.BF16_KEYS else v) for k, v in ref.items()}
    mine = ref if rank == 0 else {k: None for k in ref}  # only rank 0 knows the request: shapes travel in the header
    out = a_dist.broadcast_conditioning(mine, src=0, capacity_bytes=1 << 16)
    ok = all(out[k].dtype == want[k].dtype and torch.equal(out[k], want[k]) for k in ref)
    # the small per-request scalars (HOST_KEYS) rode INSIDE the header: CPU tensors on every rank, bit for bit (int64 seeds above 2^32,
    # fp64 knobs), no payload bytes and no second device read for them (round 4)
    ok = ok and all(out[k].device.type == "cpu" for k in ("seeds", "knobs")) and out["seeds"].tolist() == ref["seeds"].tolist()
    out2 = a_dist.broadcast_conditioning(mine, src=0, capacity_bytes=1 << 16)  # results are private copies
    ok = ok and all(torch.equal(out[k], want[k]) and torch.equal(out2[k], want[k]) for k in ref)
    for bad in ({"big": torch.zeros(1 << 15) if rank == 0 else None},                       # larger than the limit
                {f"k{i}": (torch.zeros(1) if rank == 0 else None) for i in range(17)},       # too many items
                {"deep": torch.zeros(1, 1, 1, 1, 1) if rank == 0 else None}):                # too many dimensions
        try:  # a bundle the source cannot ship raises on EVERY rank (the error rides in the header: no dead-lock)
            a_dist.broadcast_conditioning(bad, src=0, capacity_bytes=1 << 16)
            ok = False
        except ValueError:
            pass
    # run_request: broadcast -> shard -> execute -> gather, the function bench.py and NativeHandler.generate_music(data_parallel=True) call
    G5 = 5
    enc5 = torch.randn(1, 7, 16, generator=g).expand(G5, -1, -1).clone()
    enc5[3] += 1.0                                            # songs 0,1,2,4 share a caption, song 3 has its own
    ctx5 = torch.randn(1, 6, 128, generator=g)
    seeds5 = [10, 11, 12, 13, 14]
    # a source rank WITHOUT a request: the refusal rides in the header and raises on every rank (it used to raise on src before the
    # broadcast and leave the others waiting in it: advisor r3)
    try:
        a_dist.run_request(None, lambda local: None, src=0, device=torch.device("cpu"))
        ok = False
    except ValueError as e:
        ok = ok and "no request" in str(e)
    req = a_dist.pack_request(enc5, ctx5.expand(G5, -1, -1), seeds5, torch.ones(16), timesteps=[1.0, 0.6, 0.25, 0.0], inference_steps=9,
                              guidance_scale=3.5, latent_rescale=0.5, use_tiled_decode=0.0) if rank == 0 else None
    if rank == 0:
        ok = ok and tuple(req["enc_rows"].shape) == (2, 7, 16) and req["enc_index"].tolist() == [0, 0, 0, 1, 0] and req["ctx"].shape[0] == 1
    calls = []

    def execute(local):
        calls.append(local)
        return local["encoder_hidden_states"].float().sum(dim=(1, 2)).reshape(-1, 1, 1) + torch.tensor(local["seeds"], dtype=torch.float32).reshape(-1, 1, 1)

    res = a_dist.run_request(req, execute, src=0, device=torch.device("cpu"), gather=True)
    s5, e5 = a_dist.shard_range(G5, world, rank)
    loc = calls[0]
    ok = ok and res["range"] == (s5, e5) and res["global_batch"] == G5 and loc["seeds"] == seeds5[s5:e5]
    ok = ok and torch.equal(loc["encoder_hidden_states"].float(), enc5[s5:e5].to(torch.bfloat16).float())
    ok = ok and tuple(loc["context_latents"].shape) == (e5 - s5, 6, 128) and loc["knobs"]["inference_steps"] == 9.0 and loc["knobs"]["guidance_scale"] == 3.5
    ok = ok and loc["null_condition_emb"] is not None and float(loc["null_condition_emb"].float().sum()) == 16.0
    # every per-request setting is rank 0's: explicit timesteps and the decode-side scalars travel too (defaults for what was not given)
    ok = ok and loc["timesteps"] == [1.0, 0.6000000238418579, 0.25, 0.0] and loc["knobs"]["latent_rescale"] == 0.5
    ok = ok and loc["knobs"]["use_tiled_decode"] == 0.0 and loc["knobs"]["latent_shift"] == 0.0 and loc["knobs"]["shift"] == 1.0
    if rank == 0:
        allw = torch.cat(res["gathered"], 0).reshape(-1)
        expect = enc5.to(torch.bfloat16).float().sum(dim=(1, 2)) + torch.tensor(seeds5, dtype=torch.float32)
        ok = ok and torch.equal(allw, expect)
    else:
        ok = ok and res["gathered"] is None
    # a cover request (round 5, advisor r4): strengths as knobs, src_latents in fp32 bit for bit, the non-cover conditions as distinct
    # rows + index like the cover ones, a shared non-cover context collapsed to one row and expanded again on arrival
    src5 = torch.randn(G5, 6, 64, generator=g)
    enc_nc5 = torch.randn(1, 4, 16, generator=g).expand(G5, -1, -1).clone()
    enc_nc5[1] -= 2.0
    ctx_nc5 = torch.randn(1, 6, 128, generator=g)
    req = a_dist.pack_request(enc5, ctx5, seeds5, None, audio_cover_strength=0.5, cover_noise_strength=0.25, src_latents=src5,
                              encoder_hidden_states_non_cover=enc_nc5, context_latents_non_cover=ctx_nc5.expand(G5, -1, -1)) if rank == 0 else None
    if rank == 0:
        ok = ok and tuple(req["enc_rows_non_cover"].shape) == (2, 4, 16) and req["enc_index_non_cover"].tolist() == [0, 1, 0, 0, 0] and req["ctx_non_cover"].shape[0] == 1
    calls.clear()
    a_dist.run_request(req, lambda local: calls.append(local), src=0, device=torch.device("cpu"))
    loc = calls[0]
    ok = ok and loc["knobs"]["audio_cover_strength"] == 0.5 and loc["knobs"]["cover_noise_strength"] == 0.25 and loc["knobs"]["inference_steps"] == 27.0
    ok = ok and loc["src_latents"].dtype == torch.float32 and torch.equal(loc["src_latents"], src5[s5:e5])
    ok = ok and torch.equal(loc["encoder_hidden_states_non_cover"].float(), enc_nc5[s5:e5].to(torch.bfloat16).float())
    ok = ok and torch.equal(loc["context_latents_non_cover"].float(), ctx_nc5.to(torch.bfloat16).float().expand(e5 - s5, -1, -1))
    ok = ok and loc["null_condition_emb"] is None and loc["timesteps"] is None
    # ... and a request without them leaves all three None on every rank
    req = a_dist.pack_request(enc5, ctx5, seeds5) if rank == 0 else None
    calls.clear()
    a_dist.run_request(req, lambda local: calls.append(local), src=0, device=torch.device("cpu"))
    loc = calls[0]
    ok = ok and loc["src_latents"] is None and loc["encoder_hidden_states_non_cover"] is None and loc["context_latents_non_cover"] is None
    ok = ok and loc["knobs"]["audio_cover_strength"] == 1.0 and loc["knobs"]["cover_noise_strength"] == 0.0
    # NativeHandler.generate_music(data_parallel=True) on handlers that were never initialised: the failure of every rank's share
    # becomes the reference's error payload on EVERY rank (agreed by one all-reduce), nobody hangs in a collective
    from ace355.backend import NativeHandler
    hd = NativeHandler()
    pay = hd.generate_music(enc5 if rank == 0 else None, ctx5.expand(G5, -1, -1) if rank == 0 else None, seed=seeds5 if rank == 0 else None,
                            inference_steps=2, data_parallel=True)
    ok = ok and pay["success"] is False and pay["audios"] == [] and isinstance(pay["error"], str)
    pay = hd.generate_music(enc5 if rank == 0 else None, ctx5 if rank == 0 else None, seed=7, data_parallel=True)   # scalar seed: refused on rank 0, error everywhere
    ok = ok and pay["success"] is False and ("seed" in pay["error"] or "another rank" in pay["error"])
    pay = hd.generate_music(enc5 if rank == 0 else None, ctx5.expand(G5, -1, -1) if rank == 0 else None, seed=seeds5 if rank == 0 else None,
                            data_parallel=True, infer_method_typo="ode")   # a keyword that does not travel: refused, error everywhere
    ok = ok and pay["success"] is False and ("not part of the broadcast request" in pay["error"] or "another rank" in pay["error"])
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
    for scaling, want_g, want_b0 in (("weak", 16, 8), ("strong", 8, 4), (None, 8, 4)):   # default = the section-8e split of ONE batch of 8
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run",
                            "--lm-hints", "--duration", "2"] + (["--scaling", scaling] if scaling else []), capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout  # rank 0 only
        out = json.loads(lines[0])
        assert out["n_gpus"] == 2 and out["dry_run"] and out["collectives_ok"] and out["scaling"] == (scaling or "strong")
        assert out["config"]["global_batch"] == want_g and out["config"]["batch_rank0"] == want_b0
