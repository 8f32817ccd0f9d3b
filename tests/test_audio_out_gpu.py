"""Output stage on the GPU, through the C ABI: normalize_audio bit for bit against vectors of the reference's
own function (G10), the float -> PCM_16 kernel against the oracle's rule, and files written by AudioSaver read back with
independent readers (oracle RFC 9639 decoder, scipy WAV reader) at small sizes and by property (lossless round trip, MD5,
CRCs) at the metric's size (8 x 30 s stereo)."""
import hashlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_normalize_audio_bit_exact_vs_reference_vectors(gpu_device, golden_dir):
    from ace355.audio_out import normalize_audio, normalize_audio_batch
    G = np.load(f"{golden_dir}/g10_normalize_audio.npz")
    for name in G["names"].tolist():
        x = torch.from_numpy(G[f"{name}_in"]).to(gpu_device)
        for j, db in enumerate(G["dbs"].tolist()):
            got = normalize_audio(x, db)
            assert torch.equal(got.cpu(), torch.from_numpy(G[f"{name}_out{j}"])), (name, db)
            assert (got is x) == (name == "silent"), "silence is handed back untouched, everything else is a new tensor"
    names = [n for n in G["names"].tolist() if n != "mono"]
    batch = torch.stack([torch.from_numpy(G[f"{n}_in"]) for n in names]).to(gpu_device)
    out, peaks = normalize_audio_batch(batch, -3.0)
    for i, n in enumerate(names):
        assert torch.equal(out[i].cpu(), torch.from_numpy(G[f"{n}_out1"])), n
        assert float(peaks[i]) == float(np.abs(G[f"{n}_in"]).max())
    assert out.data_ptr() != batch.data_ptr()
    same, _ = normalize_audio_batch(batch, -3.0, inplace=True)
    assert same.data_ptr() == batch.data_ptr() and torch.equal(same, out)
    with pytest.raises(TypeError):
        normalize_audio(torch.zeros(2, 4))


def test_pcm16_rule_vs_oracle(gpu_device):
    from ace355.audio_out import float_to_pcm16
    from oracle import audio_out as o_audio
    g = torch.Generator().manual_seed(3)
    x = torch.rand(3, 2, 5001, generator=g) * 2 - 1
    ties = (torch.arange(-40, 40, dtype=torch.float32) + 0.5) / 32767.0           # x * 32767 lands on k + 0.5 (up to rounding)
    x[0, 0, :80] = ties
    x[1, 1, :6] = torch.tensor([1.0, -1.0, 1.5, -1.5, 32768.0 / 32767.0, 0.0])   # full scale and beyond (saturated)
    got = float_to_pcm16(x.to(gpu_device)).cpu().numpy()
    ref = o_audio.float_to_pcm16(x.numpy()).transpose(0, 2, 1)                    # [B, S, C]
    assert got.dtype == np.int16 and got.shape == (3, 5001, 2) and np.array_equal(got, ref)
    mono = float_to_pcm16(x[0, :1].to(gpu_device)).cpu().numpy()
    assert mono.shape == (5001, 1) and np.array_equal(mono[:, 0], ref[0, :, 0])


def test_saved_files_read_back(gpu_device, tmp_path):
    from scipy.io import wavfile
    from ace355.audio_out import AudioSaver, normalize_audio_batch
    from oracle import audio_out as o_audio
    g = torch.Generator().manual_seed(8)
    n = 4096 * 2 + 777
    t = torch.arange(n) / 48000.0
    wav = torch.stack([torch.stack([0.4 * torch.sin(2 * np.pi * (200 + 50 * b) * t), 0.3 * torch.sin(2 * np.pi * (200 + 50 * b) * t + 0.2)]) for b in range(3)])
    wav = wav + 0.01 * torch.randn(3, 2, n, generator=g)
    dev = wav.to(gpu_device)
    norm, _ = normalize_audio_batch(dev, -1.0)
    saver = AudioSaver()
    paths = saver.save_batch(norm, tmp_path / "out", file_prefix="song")
    assert [p.rsplit("/", 1)[1] for p in paths] == ["song_0000.flac", "song_0001.flac", "song_0002.flac"]
    for b, p in enumerate(paths):
        pcm, info = o_audio.flac_decode(open(p, "rb").read())                    # independent reader: CRCs, numbering, MD5
        want = o_audio.float_to_pcm16(o_audio.normalize_audio(wav[b], -1.0).numpy()).T
        assert np.array_equal(pcm, want) and info["sample_rate"] == 48000
        assert abs(int(np.abs(pcm).max()) - round(32767 * 10 ** (-1 / 20))) <= 1   # -1 dBFS peak
    for fmt in ("wav", "wav32"):
        paths = saver.save_batch(norm, tmp_path / fmt, format=fmt)
        assert all(p.endswith(".wav") for p in paths)
        for b, p in enumerate(paths):
            sr, got = wavfile.read(p)
            assert sr == 48000 and got.dtype == np.float32 and np.array_equal(got, norm[b].cpu().numpy().T)
    one = saver.save_audio(norm[1], tmp_path / "single", sample_rate=44100)
    assert one.endswith("single.flac")
    pcm, info = o_audio.flac_decode(open(one, "rb").read())
    assert info["sample_rate"] == 44100 and np.array_equal(pcm, o_audio.float_to_pcm16(norm[1].cpu().numpy()).T)
    # host tensors and [samples, channels] layouts are staged through the same path
    p2 = saver.save_audio(norm[1].cpu().T.contiguous(), tmp_path / "t", channels_first=False)
    assert open(p2, "rb").read() == open(saver.save_audio(norm[1], tmp_path / "t2"), "rb").read()
    with pytest.raises(NotImplementedError, match="ffmpeg"):
        saver.save_audio(norm[0], tmp_path / "x", format="mp3")
    with pytest.raises(RuntimeError, match="cannot write"):
        saver.save_paths(norm, [tmp_path / "no_such_dir" / f"{i}.flac" for i in range(3)])


def test_full_size_batch_round_trip(gpu_device, tmp_path):
    """8 x 30 s stereo (1 440 000 samples per channel, the metric's batch): every file decodes (native decoder: all CRCs, MD5)
    to exactly the GPU-quantised PCM; the MD5 in STREAMINFO equals hashlib's over that PCM; thread count does not change a byte."""
    import time
    from ace355.audio_out import AudioSaver, flac_decode_pcm16, float_to_pcm16, normalize_audio_batch
    B, S = 8, 1920 * 750
    g = torch.Generator(device=gpu_device).manual_seed(1)
    t = torch.arange(S, device=gpu_device) / 48000.0
    base = torch.stack([torch.sin(2 * np.pi * (110.0 * (b + 1)) * t) * (0.6 + 0.4 * torch.sin(2 * np.pi * 0.25 * t)) for b in range(B)])
    wav = torch.stack([base, 0.7 * base.roll(7, dims=1)], dim=1) + 0.02 * torch.randn(B, 2, S, device=gpu_device, generator=g)
    norm, peaks = normalize_audio_batch(wav, -1.0)
    assert torch.allclose(norm.abs().amax(dim=(1, 2)).cpu(), torch.full((B,), 10 ** (-1 / 20)), atol=1e-6)
    pcm = float_to_pcm16(norm).cpu().numpy()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    paths = AudioSaver(n_threads=16).save_paths(norm, [tmp_path / f"{i}.flac" for i in range(B)])
    dt = time.perf_counter() - t0
    sizes = []
    for b, p in enumerate(paths):
        data = open(p, "rb").read()
        got, sr = flac_decode_pcm16(data)
        assert sr == 48000 and np.array_equal(got, pcm[b]), b
        assert data[26:42] == hashlib.md5(pcm[b].tobytes()).digest()
        sizes.append(len(data))
    one = AudioSaver(n_threads=1).save_paths(norm[:1], [tmp_path / "single_thread.flac"])
    assert open(one[0], "rb").read() == open(paths[0], "rb").read()
    print(f"save 8 x 30 s as FLAC: {dt * 1e3:.1f} ms ({B / dt:.1f} songs/s), {np.mean(sizes) / pcm[0].nbytes:.3f} of the PCM size")
