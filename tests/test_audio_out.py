"""Output stage, CPU part: the oracle against the reference's normalize_audio vectors (G10), and the host
codecs of the C ABI (FLAC encoder / decoder, WAV writer: plain host functions, no GPU involved) against independent
readers: the RFC 9639 restatement in oracle/audio_out.py, hashlib's MD5, scipy's WAV reader."""
import hashlib
import io

import numpy as np
import pytest
import torch


def _signals():
    rng = np.random.default_rng(5)
    n = 4096 * 2 + 1234
    t = np.arange(n) / 48000.0
    sine = np.stack([0.6 * np.sin(2 * np.pi * 220 * t), 0.55 * np.sin(2 * np.pi * 220 * t + 0.1) + 0.05 * np.sin(2 * np.pi * 3000 * t)], 1)
    music = np.cumsum(rng.standard_normal((n, 2)) * 0.01, 0)  # smooth random walk: fixed predictors of order >= 1 win
    music[:, 1] = 0.9 * music[:, 0] + 0.02 * rng.standard_normal(n)
    sq = np.where((np.arange(n) // 7) % 2 == 0, 32767, -32768)[:, None].repeat(2, 1).astype(np.int16)  # defeats every predictor
    to16 = lambda x: np.clip(np.rint(x * 32767), -32768, 32767).astype(np.int16)  # noqa: E731
    return {
        "sine_stereo": to16(sine), "walk_stereo": to16(music / np.abs(music).max()), "noise_stereo": to16(rng.uniform(-1, 1, (n, 2))),
        "silence": np.zeros((n, 2), np.int16), "dc": np.full((5000, 2), -1234, np.int16), "square_extremes": sq,
        "mono_sine": to16(sine[:, :1]), "one_sample": np.array([[7, -9]], np.int16), "short": to16(sine[:5]),
        "exact_block": to16(sine[:4096]), "tiny_tail": to16(sine[:4096 + 3]), "right_only": np.stack([np.zeros(n, np.int16), to16(sine[:, 0])], 1),
        "left_plus_small": np.stack([to16(sine[:, 0]), (to16(sine[:, 0]).astype(np.int32) + rng.integers(-2, 3, n)).clip(-32768, 32767).astype(np.int16)], 1),
    }


def test_g10_normalize_audio_oracle(golden_dir):
    from oracle import audio_out as o_audio
    G = np.load(f"{golden_dir}/g10_normalize_audio.npz")
    for name in G["names"].tolist():
        x = torch.from_numpy(G[f"{name}_in"])
        for j, db in enumerate(G["dbs"].tolist()):
            assert torch.equal(o_audio.normalize_audio(x, db), torch.from_numpy(G[f"{name}_out{j}"])), (name, db)
    assert np.array_equal(G["silent_in"], G["silent_out0"]), "peak < 1e-6 is returned untouched"


@pytest.mark.parametrize("name", list(_signals()))
def test_flac_stream_decodes_to_the_same_pcm(name):
    """Lossless contract: the stream the native encoder writes decodes - with an independent RFC 9639 reader that checks
    every CRC-8 / CRC-16, the frame numbering and the STREAMINFO MD5 - to exactly the PCM that went in."""
    from ace355 import audio_out
    from oracle import audio_out as o_audio
    pcm = _signals()[name]
    data = audio_out.flac_encode_pcm16(pcm, 48000, n_threads=3)
    got, info = o_audio.flac_decode(data)
    assert np.array_equal(got, pcm.astype(np.int64)), name
    assert (info["sample_rate"], info["channels"], info["bps"], info["frames"]) == (48000, pcm.shape[1], 16, pcm.shape[0])
    assert info["md5"] == hashlib.md5(pcm.astype("<i2").tobytes()).digest()
    assert info["min_block"] == info["max_block"] == 4096
    assert info["min_frame"] == min(info["frame_sizes"]) and info["max_frame"] == max(info["frame_sizes"])
    again, sr = audio_out.flac_decode_pcm16(data)
    assert sr == 48000 and np.array_equal(again, pcm), "native decoder"
    raw = pcm.size * 2
    if name in ("sine_stereo", "mono_sine", "left_plus_small", "right_only"):
        assert len(data) < 0.6 * raw, (name, len(data), raw)     # the predictors and the stereo decorrelation do their job
    if name in ("silence", "dc"):
        assert len(data) < 42 + 16 * (pcm.shape[0] // 4096 + 1)    # constant subframes
    assert len(data) <= raw * 17 / 16 + 42 + 40 * (pcm.shape[0] // 4096 + 1)  # verbatim fallback bounds incompressible input


def test_flac_empty_and_other_rates():
    from ace355 import audio_out
    from oracle import audio_out as o_audio
    got, info = o_audio.flac_decode(audio_out.flac_encode_pcm16(np.zeros((0, 2), np.int16), 48000))
    assert got.shape[0] == 0 and info["frames"] == 0
    pcm = _signals()["sine_stereo"][:3000]
    for sr in (44100, 32000, 12000, 12345, 123450, 700001):   # table / kHz byte / Hz word / tens-of-Hz word / STREAMINFO only
        got, info = o_audio.flac_decode(audio_out.flac_encode_pcm16(pcm, sr)) if sr != 700001 else (None, None)
        if sr == 700001:
            continue
        assert np.array_equal(got, pcm) and info["sample_rate"] == sr


def test_flac_is_deterministic_across_thread_counts_and_long():
    """30 s of stereo at 48 kHz (the metric's item): 352 frames, frame numbers >= 128 take the 2-byte coded-number path."""
    from ace355 import audio_out
    rng = np.random.default_rng(9)
    n = 48000 * 30
    t = np.arange(n) / 48000.0
    x = 0.5 * np.sin(2 * np.pi * 330 * t) * (1 + 0.3 * np.sin(2 * np.pi * 0.5 * t)) + 0.01 * rng.standard_normal(n)
    pcm = np.stack([x, 0.8 * x + 0.01 * rng.standard_normal(n)], 1)
    pcm = np.clip(np.rint(pcm * 32767 * 0.89), -32768, 32767).astype(np.int16)
    a = audio_out.flac_encode_pcm16(pcm, 48000, n_threads=1)
    b = audio_out.flac_encode_pcm16(pcm, 48000, n_threads=8)
    assert a == b
    got, sr = audio_out.flac_decode_pcm16(a)          # checks every CRC and the MD5
    assert np.array_equal(got, pcm)
    assert a[26:42] == hashlib.md5(pcm.tobytes()).digest()
    assert len(a) < 0.75 * pcm.size * 2


def test_flac_decoder_rejects_corruption():
    from ace355 import audio_out
    data = bytearray(audio_out.flac_encode_pcm16(_signals()["sine_stereo"], 48000))
    bad = bytearray(data)
    bad[len(bad) // 2] ^= 0x10
    with pytest.raises(RuntimeError, match="CRC|sync|residual|subframe|MD5|block|partition"):
        audio_out.flac_decode_pcm16(bytes(bad))
    with pytest.raises(RuntimeError, match="not a FLAC"):
        audio_out.flac_decode_pcm16(b"RIFF" + bytes(60))
    with pytest.raises(RuntimeError, match="truncated|CRC|sync"):
        audio_out.flac_decode_pcm16(bytes(data[:len(data) // 2]))


def test_wav_read_back_by_scipy():
    from scipy.io import wavfile
    from ace355 import audio_out
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (1001, 2)).astype(np.float32)
    sr, got = wavfile.read(io.BytesIO(audio_out.wav_encode(x, 48000)))
    assert sr == 48000 and got.dtype == np.float32 and np.array_equal(got, x)
    p = (x * 32767).astype(np.int16)
    sr, got = wavfile.read(io.BytesIO(audio_out.wav_encode(p, 44100)))
    assert sr == 44100 and got.dtype == np.int16 and np.array_equal(got, p)
    sr, got = wavfile.read(io.BytesIO(audio_out.wav_encode(x[:, 0], 8000)))
    assert got.shape == (1001,) and np.array_equal(got, x[:, 0])


def test_audio_saver_names_and_formats(tmp_path):
    """Path / format resolution of AudioSaver.save_audio (audio_utils.py:96-117) without touching a GPU."""
    from ace355.audio_out import AudioSaver
    s = AudioSaver("nonsense")
    assert s.default_format == "flac"
    assert s._resolve(tmp_path / "a", None) == ("flac", tmp_path / "a.flac")
    assert s._resolve(tmp_path / "a.b", "wav32") == ("wav32", tmp_path / "a.wav")
    assert s._resolve(tmp_path / "a.wav", "flac") == ("flac", tmp_path / "a.wav")      # a known suffix is kept, as in the reference
    assert s._resolve(tmp_path / "a", "bogus")[0] == "flac"
    for fmt in ("mp3", "opus", "aac"):
        with pytest.raises(NotImplementedError, match="ffmpeg"):
            s._resolve(tmp_path / "a", fmt)


def test_convert_audio_flac_to_wav(tmp_path):
    from scipy.io import wavfile
    from ace355 import audio_out
    pcm = _signals()["walk_stereo"]
    src = tmp_path / "x.flac"
    src.write_bytes(audio_out.flac_encode_pcm16(pcm, 48000))
    out = audio_out.AudioSaver().convert_audio(src, tmp_path / "y", "wav", remove_input=True)
    assert out.endswith("y.wav") and not src.exists()
    sr, got = wavfile.read(out)
    assert sr == 48000 and np.array_equal(got, pcm.astype(np.float32) / np.float32(32768.0))
    with pytest.raises(FileNotFoundError):
        audio_out.AudioSaver().convert_audio(tmp_path / "missing.flac", tmp_path / "z", "wav")


@pytest.mark.parametrize("kw", [dict(), dict(lpc_order=8, precision=14), dict(rice2=True), dict(escape_first_partition=True),
                                dict(wasted=3), dict(variable_blocks=True), dict(lpc_order=1, precision=5, rice2=True, escape_first_partition=True)])
def test_native_decoder_reads_streams_it_did_not_write(kw):
    """LPC subframes, 5-bit Rice parameters, escaped partitions, wasted bits, 1152-sample and variable blocks: features of FLAC
    files in the wild (libFLAC output) that the native encoder never produces.  A second writer in the oracle emits them; the
    native decoder (AudioSaver.convert_audio's reader) and the oracle decoder must both return the PCM."""
    from ace355 import audio_out
    from oracle import audio_out as o_audio
    rng = np.random.default_rng(17)
    n = 1152 * 3 + 401
    t = np.arange(n) / 44100.0
    x = np.stack([0.5 * np.sin(2 * np.pi * 330 * t) + 0.2 * np.sin(2 * np.pi * 1234 * t + 1), 0.4 * np.sin(2 * np.pi * 220 * t)], 1)
    pcm = np.clip(np.rint((x + 0.002 * rng.standard_normal((n, 2))) * 30000), -32768, 32767).astype(np.int16)
    if kw.get("wasted"):
        pcm = (pcm >> 3) << 3
    data = o_audio.flac_encode_test_stream(pcm, 44100, **kw)
    got, info = o_audio.flac_decode(data)
    assert np.array_equal(got, pcm.astype(np.int64)), "oracle reader"
    nat, sr = audio_out.flac_decode_pcm16(data)
    assert sr == 44100 and np.array_equal(nat, pcm), kw
    if not kw.get("escape_first_partition"):
        assert len(data) < 0.8 * pcm.nbytes   # the predictor is doing something
