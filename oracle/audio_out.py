"""CPU restatement of the output stage (test oracle).

Reference: acestep/audio_utils.py:24-62 (``normalize_audio``), :65-215 (``AudioSaver.save_audio``: torchaudio ->
libsndfile -> libFLAC for "flac", IEEE-float RIFF/WAVE for "wav"/"wav32"), acestep/inference.py:649-726 (call order).

Pinning status:
  * ``normalize_audio`` is PINNED: tests/golden/make_golden.py (fixture G10) executes the reference's own function body
    (extracted from audio_utils.py with ``ast`` because the module imports the absent ``torchaudio``) and this
    restatement must reproduce it bit for bit.
  * ``float_to_pcm16`` is **parity unpinned**: the rule lives in libsndfile (``f2flac16_array`` / ``f2les_array``:
    ``lrintf(x * 0x7FFF)`` when float normalisation is on and clipping is off, the library defaults), a C dependency of the
    third-party ``soundfile`` package; neither is in /root/reference or in this image.
  * ``flac_decode`` restates the FLAC format from RFC 9639 (sections 9-11: frame header, subframes, residual coding): an
    independent reader for the streams the native encoder writes.  **Parity unpinned** against libFLAC for the same
    reason; FLAC being lossless, the testable contract is decode(encode(pcm)) == pcm plus the STREAMINFO MD5 (hashlib).
  * WAV files are read back with ``scipy.io.wavfile`` (an independent implementation present in the image).

Test infrastructure only: nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import hashlib
from typing import Dict, Tuple

import numpy as np
import torch


def normalize_audio(audio: torch.Tensor, target_db: float = -1.0) -> torch.Tensor:
    """audio_utils.py:24-62: peak over the whole tensor; untouched below 1e-6; gain = 10^(dB/20) / peak."""
    peak = torch.max(torch.abs(audio))
    if peak < 1e-6:
        return audio
    gain = (10 ** (target_db / 20.0)) / peak
    return audio.clone() * gain


def float_to_pcm16(audio: np.ndarray) -> np.ndarray:
    """libsndfile float -> PCM_16 (normalisation on, clipping off): lrintf(x * 32767), round half to even; saturated here
    (libsndfile would wrap; the normalised signal never gets there)."""
    v = np.rint(audio.astype(np.float32) * np.float32(32767.0))
    return np.clip(v, -32768, 32767).astype(np.int16)


# ---------------------------------------------------------------------------------------------------- FLAC (RFC 9639)
class _Bits:
    def __init__(self, data: bytes, pos: int = 0):
        self.d, self.p = data, pos * 8

    def get(self, n: int) -> int:
        v = 0
        while n > 0:
            byte = self.d[self.p >> 3]
            avail = 8 - (self.p & 7)
            take = min(n, avail)
            v = (v << take) | ((byte >> (avail - take)) & ((1 << take) - 1))
            self.p += take
            n -= take
        return v

    def sget(self, n: int) -> int:
        v = self.get(n)
        return v - (1 << n) if n and v >> (n - 1) else v

    def unary(self) -> int:
        q = 0
        while self.get(1) == 0:
            q += 1
        return q

    def align(self):
        self.p = (self.p + 7) & ~7


def _crc(data: bytes, poly: int, width: int) -> int:
    top, mask, c = 1 << (width - 1), (1 << width) - 1, 0
    for b in data:
        c ^= b << (width - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(br: _Bits, n: int, bps: int) -> list:
    assert br.get(1) == 0, "subframe padding"
    typ = br.get(6)
    wasted = 0
    if br.get(1):
        wasted = 1 + br.unary()
    bps -= wasted
    if typ == 0:
        out = [br.sget(bps)] * n
    elif typ == 1:
        out = [br.sget(bps) for _ in range(n)]
    else:
        if 8 <= typ <= 12:
            order, coefs, shift = typ & 7, None, 0
        elif typ >= 32:
            order = (typ & 31) + 1
            coefs = None
        else:
            raise ValueError(f"reserved subframe type {typ}")
        out = [br.sget(bps) for _ in range(order)]
        if typ >= 32:
            prec = br.get(4) + 1
            shift = br.sget(5)
            coefs = [br.sget(prec) for _ in range(order)]
        else:
            coefs = _FIXED[order]
        method = br.get(2)
        assert method in (0, 1), "residual coding method"
        kbits, esc = (4, 15) if method == 0 else (5, 31)
        po = br.get(4)
        assert n % (1 << po) == 0 or po == 0
        plen = n >> po
        res = []
        for p in range(1 << po):
            cnt = plen - (order if p == 0 else 0)
            k = br.get(kbits)
            if k == esc:
                raw = br.get(5)
                res.extend(br.sget(raw) for _ in range(cnt))
            else:
                for _ in range(cnt):
                    u = (br.unary() << k) | br.get(k)
                    res.append((u >> 1) ^ -(u & 1))
        for r in res:
            pred = sum(c * out[-1 - j] for j, c in enumerate(coefs)) >> shift
            out.append(r + pred)
    return [v << wasted for v in out] if wasted else out


def flac_decode(data: bytes) -> Tuple[np.ndarray, Dict]:
    """-> (int array [frames, channels], info).  Checks every CRC-8 / CRC-16 and the STREAMINFO MD5 (16-bit streams)."""
    assert data[:4] == b"fLaC", "marker"
    pos, info = 4, None
    while True:
        last, typ = data[pos] >> 7, data[pos] & 0x7F
        ln = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + ln]
        if typ == 0:
            br = _Bits(body)
            info = {"min_block": br.get(16), "max_block": br.get(16), "min_frame": br.get(24), "max_frame": br.get(24),
                    "sample_rate": br.get(20), "channels": br.get(3) + 1, "bps": br.get(5) + 1, "frames": br.get(36),
                    "md5": bytes(body[18:34])}
        pos += 4 + ln
        if last:
            break
    assert info is not None, "STREAMINFO"
    C, chans, done, frame_no, sizes = info["channels"], [[] for _ in range(info["channels"])], 0, 0, []
    while done < info["frames"]:
        start = pos
        br = _Bits(data, pos)
        assert br.get(14) == 0x3FFE and br.get(1) == 0, "sync"
        variable = br.get(1)
        bs_code, sr_code, assign, ss_code = br.get(4), br.get(4), br.get(4), br.get(3)
        assert br.get(1) == 0
        first = br.get(8)
        nb = 0
        while first & (0x80 >> nb):
            nb += 1
        num = first & (0x7F >> nb)
        for _ in range(max(nb - 1, 0)):
            cont = br.get(8)
            assert cont >> 6 == 2
            num = (num << 6) | (cont & 0x3F)
        if not variable:
            assert num == frame_no, "frame numbers count up from 0"
        if bs_code == 1:
            bs = 192
        elif 2 <= bs_code <= 5:
            bs = 576 << (bs_code - 2)
        elif bs_code == 6:
            bs = br.get(8) + 1
        elif bs_code == 7:
            bs = br.get(16) + 1
        else:
            assert bs_code >= 8
            bs = 256 << (bs_code - 8)
        rates = {1: 88200, 2: 176400, 3: 192000, 4: 8000, 5: 16000, 6: 22050, 7: 24000, 8: 32000, 9: 44100, 10: 48000, 11: 96000}
        if sr_code == 0:
            sr = info["sample_rate"]
        elif sr_code in rates:
            sr = rates[sr_code]
        elif sr_code == 12:
            sr = br.get(8) * 1000
        elif sr_code == 13:
            sr = br.get(16)
        else:
            assert sr_code == 14
            sr = br.get(16) * 10
        assert sr == info["sample_rate"]
        bps = {0: info["bps"], 1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}[ss_code]
        hdr_end = br.p >> 3
        assert _crc(data[start:hdr_end], 0x07, 8) == br.get(8), "CRC-8"
        nch = assign + 1 if assign < 8 else 2
        assert assign <= 10 and nch == C
        sub = []
        for c in range(nch):
            side = (assign == 8 and c == 1) or (assign == 9 and c == 0) or (assign == 10 and c == 1)
            sub.append(_subframe(br, bs, bps + (1 if side else 0)))
        br.align()
        end = br.p >> 3
        assert _crc(data[start:end], 0x8005, 16) == br.get(16), "CRC-16"
        if assign == 8:
            sub = [sub[0], [a - b for a, b in zip(sub[0], sub[1])]]
        elif assign == 9:
            sub = [[a + b for a, b in zip(sub[0], sub[1])], sub[1]]
        elif assign == 10:
            L, R = [], []
            for m, s in zip(sub[0], sub[1]):
                m = (m << 1) | (s & 1)
                L.append((m + s) >> 1)
                R.append((m - s) >> 1)
            sub = [L, R]
        for c in range(nch):
            chans[c].extend(sub[c])
        pos = br.p >> 3
        sizes.append(pos - start)
        done += bs
        frame_no += 1
    pcm = np.array(chans, dtype=np.int64).T.copy()
    info["frame_sizes"] = sizes
    if info["bps"] == 16 and info["md5"] != bytes(16):
        assert hashlib.md5(pcm.astype("<i2").tobytes()).digest() == info["md5"], "STREAMINFO MD5"
    return pcm, info


# ---------------------------------------------------------------------------------------------------- test-stream writer
class _BitsOut:
    def __init__(self):
        self.b, self.acc, self.n = bytearray(), 0, 0

    def put(self, v: int, n: int):
        if n == 0:
            return
        self.acc = (self.acc << n) | (v & ((1 << n) - 1))
        self.n += n
        while self.n >= 8:
            self.n -= 8
            self.b.append((self.acc >> self.n) & 0xFF)
        self.acc &= (1 << self.n) - 1

    def align(self):
        if self.n:
            self.put(0, 8 - self.n)


def flac_encode_test_stream(pcm: np.ndarray, sample_rate: int = 44100, block: int = 1152, lpc_order: int = 4, precision: int = 12,
                            rice2: bool = False, escape_first_partition: bool = False, wasted: int = 0, variable_blocks: bool = False) -> bytes:
    """A deliberately different FLAC writer (RFC 9639 features the native encoder never emits: LPC subframes with quantised
    coefficients, 5-bit Rice parameters, an escaped partition, wasted bits, 1152-sample blocks, sample-number frame headers) so
    that the native DECODER is exercised on streams it did not write.  pcm int [frames, channels] (independent channels)."""
    pcm = np.asarray(pcm, dtype=np.int64)
    frames, ch = pcm.shape
    bps = 16
    out = bytearray(b"fLaC")
    si = _BitsOut()
    si.put(0x80, 8), si.put(34, 24)
    si.put(block if not variable_blocks else 16, 16), si.put(block, 16), si.put(0, 24), si.put(0, 24)
    si.put(sample_rate, 20), si.put(ch - 1, 3), si.put(bps - 1, 5), si.put(frames, 36)
    for byte in hashlib.md5(pcm.astype("<i2").tobytes()).digest():
        si.put(byte, 8)
    out += si.b
    pos, fno = 0, 0
    while pos < frames:
        n = min(block if not variable_blocks else max(16, block >> (fno % 3)), frames - pos)
        fr = _BitsOut()
        fr.put(0x3FFE, 14), fr.put(0, 1), fr.put(1 if variable_blocks else 0, 1)
        fr.put(7, 4)                                     # 16-bit block size at the end of the header
        fr.put({44100: 9, 48000: 10}.get(sample_rate, 0), 4)
        fr.put(ch - 1, 4), fr.put(4, 3), fr.put(0, 1)
        num = pos if variable_blocks else fno            # UTF-8-like coded number
        if num < 0x80:
            fr.put(num, 8)
        else:
            nb = 2
            while nb < 7 and (num >> (5 * nb + 1)):
                nb += 1
            fr.put(((0xFF00 >> nb) & 0xFF) | (num >> (6 * (nb - 1))), 8)
            for i in range(nb - 2, -1, -1):
                fr.put(0x80 | ((num >> (6 * i)) & 0x3F), 8)
        fr.put(n - 1, 16)
        fr.put(_crc(bytes(fr.b), 0x07, 8), 8)
        for c in range(ch):
            s = pcm[pos:pos + n, c].copy()
            w = wasted if wasted and np.all((s & ((1 << wasted) - 1)) == 0) else 0
            s >>= w
            b = bps - w
            order = min(lpc_order, n - 1)
            fr.put(0, 1), fr.put(32 | (order - 1), 6)
            if w:
                fr.put(1, 1)
                for _ in range(w - 1):
                    fr.put(0, 1)
                fr.put(1, 1)
            else:
                fr.put(0, 1)
            # least-squares predictor on this block, quantised to `precision` bits with a common shift
            X = np.stack([s[order - 1 - k:n - 1 - k] for k in range(order)], 1).astype(np.float64)
            coef = np.linalg.lstsq(X, s[order:].astype(np.float64), rcond=None)[0] if n > order else np.zeros(order)
            cmax = max(np.abs(coef).max(), 1e-9)
            shift = int(max(0, min(15, precision - 1 - int(np.ceil(np.log2(cmax + 1e-12))) - 1)))
            q = np.clip(np.rint(coef * (1 << shift)), -(1 << (precision - 1)), (1 << (precision - 1)) - 1).astype(np.int64)
            for k in range(order):
                fr.put(int(s[k]), b)
            fr.put(precision - 1, 4), fr.put(shift, 5)
            for k in range(order):
                fr.put(int(q[k]), precision)
            pred = (X.astype(np.int64) @ q) >> shift if n > order else np.zeros(0, np.int64)
            res = s[order:] - pred
            fr.put(1 if rice2 else 0, 2)
            po = 0
            while po < 3 and (n >> (po + 1)) << (po + 1) == n and (n >> (po + 1)) > order:
                po += 1
            fr.put(po, 4)
            plen, i0 = n >> po, 0
            for p in range(1 << po):
                cnt = plen - (order if p == 0 else 0)
                r = res[i0:i0 + cnt]
                i0 += cnt
                u = np.where(r >= 0, 2 * r, -2 * r - 1)
                if escape_first_partition and p == 0:
                    fr.put(31 if rice2 else 15, 5 if rice2 else 4)
                    nb = int(max(1, int(np.abs(r).max(initial=0)))).bit_length() + 1
                    fr.put(nb, 5)
                    for v in r:
                        fr.put(int(v), nb)
                    continue
                k = int(max(0, int(np.log2(max(u.mean() if cnt else 1, 1)))))
                k = min(k, 30 if rice2 else 14)
                fr.put(k, 5 if rice2 else 4)
                for v in u:
                    v = int(v)
                    for _ in range(v >> k):
                        fr.put(0, 1)
                    fr.put(1, 1)
                    fr.put(v, k)
        fr.align()
        crc = _crc(bytes(fr.b), 0x8005, 16)
        out += fr.b
        out += bytes([crc >> 8, crc & 0xFF])
        pos += n
        fno += 1
    return bytes(out)
