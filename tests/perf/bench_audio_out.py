#!/usr/bin/env python
"""Output-stage timing at the metric's batch: 8 decoded 30 s stereo songs in HBM -> normalised files.
GPU path: normalize_audio_batch + AudioSaver.save_paths (GPU quantise/interleave, one D2H copy, pooled FLAC frame jobs).
CPU baseline ("port"): the reference's loop shape (inference.py:649-726) - per item .cpu(), oracle normalize_audio on the
host, numpy quantise, then THIS library's FLAC encoder on one thread (libFLAC / libsndfile / torchaudio are absent, so the
codec itself cannot be the reference's; the loop structure is).
Usage: python tests/perf/bench_audio_out.py [--batch 8] [--seconds 30] [--dir /dev/shm]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ace355.audio_out import AudioSaver, flac_encode_pcm16, normalize_audio_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--dir", default="/dev/shm" if os.path.isdir("/dev/shm") else None)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B, S = args.batch, int(args.seconds * 48000)
    g = torch.Generator(device=dev).manual_seed(1)
    t = torch.arange(S, device=dev) / 48000.0
    base = torch.stack([sum(torch.sin(2 * np.pi * (110.0 * (b + 1) * h) * t) / h for h in (1, 2, 3, 5)) * (0.6 + 0.4 * torch.sin(2 * np.pi * 0.25 * t))
                        for b in range(B)])
    wav = torch.stack([base, 0.7 * base.roll(7, dims=1)], dim=1) + 0.05 * torch.randn(B, 2, S, device=dev, generator=g)
    out_dir = tempfile.mkdtemp(dir=args.dir)
    res = {"workload": f"{B} x {args.seconds:g} s stereo 48 kHz fp32 in HBM -> normalize_audio(-1 dB) -> files", "dir": out_dir}
    try:
        for fmt, threads in (("flac", 1), ("flac", 4), ("flac", 16), ("flac", 32), ("wav", 16)):
            saver = AudioSaver(n_threads=threads)
            paths = [os.path.join(out_dir, f"{fmt}{threads}_{i}") for i in range(B)]
            for it in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                norm, _ = normalize_audio_batch(wav, -1.0)
                written = saver.save_paths(norm, paths, format=fmt)
                dt = time.perf_counter() - t0
            res[f"{fmt}_{threads}thr_ms"] = round(dt * 1e3, 2)
            res[f"{fmt}_{threads}thr_songs_per_s"] = round(B / dt, 1)
            if fmt == "flac":
                res["flac_ratio"] = round(sum(os.path.getsize(p) for p in written) / (B * S * 4), 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            norm, _ = normalize_audio_batch(wav, -1.0)
        torch.cuda.synchronize()
        res["gpu_normalize_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
        if not args.no_cpu:
            from oracle import audio_out as o_audio
            torch.set_num_threads(1)
            n_items = min(B, 2)
            t0 = time.perf_counter()
            for b in range(n_items):
                x = wav[b].cpu()
                x = o_audio.normalize_audio(x, -1.0)
                pcm = o_audio.float_to_pcm16(x.numpy()).T.copy()
                data = flac_encode_pcm16(pcm, 48000, n_threads=1)
                with open(os.path.join(out_dir, f"cpu_{b}.flac"), "wb") as f:
                    f.write(data)
            cpu = (time.perf_counter() - t0) / n_items
            res.update({"cpu_loop_ms_per_song": round(cpu * 1e3, 1), "cpu_loop_songs_per_s": round(1 / cpu, 1), "cpu_sample": f"{n_items} items, 1 thread"})
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
