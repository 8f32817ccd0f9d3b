// audio_out.hip - SURVEY.md 8f row N4, the output stage after the VAE decode:
//   acestep/inference.py:649-726 (per item: normalize_audio, save_audio) and acestep/audio_utils.py:24-62, 65-215.
// The reference walks the batch in one Python thread: .cpu() of the fp32 waveform, three passes of normalize_audio on the
// host, then torchaudio -> libsndfile -> libFLAC.  Here the sample-level work stays on the GPU (absmax + gain, float ->
// PCM_16 with channel interleave: 11.5 MB fp32 per 30 s song leave HBM as 5.8 MB of int16), and the byte-level work
// (FLAC frames are independent: fixed predictors + partitioned Rice coding per 4096-sample block, CRC-8/16, MD5 of the
// PCM for STREAMINFO) runs on a pool of host threads over (item, block) jobs.  Containers follow RFC 9639 (FLAC) and the
// RIFF/WAVE layout; torchaudio / libsndfile / libFLAC are absent from the reference tree and from this image, so byte
// equality with their output is not claimed - sample equality after decoding is (FLAC is lossless).
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ace355.h"
#include "common.h"

namespace ace355 {
namespace {

constexpr int kBlock = 4096;  // FLAC block size (code 1100)

// ------------------------------------------------------------------------------------------------ bit I/O, CRCs, MD5
struct Crc {
    uint8_t t8[256];
    uint16_t t16[256];
    Crc() {
        for (int i = 0; i < 256; ++i) {
            uint8_t c = (uint8_t)i;
            for (int b = 0; b < 8; ++b) c = (uint8_t)((c & 0x80) ? ((c << 1) ^ 0x07) : (c << 1));
            t8[i] = c;
            uint16_t d = (uint16_t)(i << 8);
            for (int b = 0; b < 8; ++b) d = (uint16_t)((d & 0x8000) ? ((d << 1) ^ 0x8005) : (d << 1));
            t16[i] = d;
        }
    }
    uint8_t crc8(const uint8_t* p, size_t n) const {
        uint8_t c = 0;
        for (size_t i = 0; i < n; ++i) c = t8[c ^ p[i]];
        return c;
    }
    uint16_t crc16(const uint8_t* p, size_t n) const {
        uint16_t c = 0;
        for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ t16[(c >> 8) ^ p[i]]);
        return c;
    }
};
const Crc& crc() {
    static const Crc c;
    return c;
}

// Writes into a caller-sized buffer through a register-resident accumulator (a std::vector reference here costs a store
// to the vector header per byte, and neighbouring frames' headers share cache lines across the worker threads).
struct BitWriter {
    uint8_t* base;
    uint8_t* p;
    uint64_t acc = 0;
    int nbits = 0;
    explicit BitWriter(uint8_t* b) : base(b), p(b) {}
    void put(uint32_t v, int n) {  // n in [0, 32], MSB first
        if (n == 0) return;
        acc = (acc << n) | (n == 32 ? (uint64_t)v : (uint64_t)(v & ((1u << n) - 1u)));
        nbits += n;
        if (nbits >= 32) {
            nbits -= 32;
            const uint32_t w = (uint32_t)(acc >> nbits);
            p[0] = (uint8_t)(w >> 24), p[1] = (uint8_t)(w >> 16), p[2] = (uint8_t)(w >> 8), p[3] = (uint8_t)w;
            p += 4;
        }
    }
    void put_signed(int32_t v, int n) { put((uint32_t)v, n); }
    void unary(uint32_t q) {  // q zero bits, then a one
        while (q >= 32) {
            put(0, 32);
            q -= 32;
        }
        put(1, (int)q + 1);
    }
    void align() {
        if (nbits & 7) put(0, 8 - (nbits & 7));
    }
    size_t flush() {  // at a byte boundary: push the pending whole bytes out, return the size so far
        while (nbits >= 8) {
            nbits -= 8;
            *p++ = (uint8_t)(acc >> nbits);
        }
        return (size_t)(p - base);
    }
};

struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;  // pos in bits
    bool bad = false;
    BitReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
    uint32_t get(int k) {  // k in [0, 32]
        uint32_t v = 0;
        while (k > 0) {
            if ((pos >> 3) >= n) {
                bad = true;
                return 0;
            }
            const int avail = 8 - (int)(pos & 7);
            const int take = k < avail ? k : avail;
            const uint32_t byte = p[pos >> 3];
            v = (v << take) | ((byte >> (avail - take)) & ((1u << take) - 1u));
            pos += take;
            k -= take;
        }
        return v;
    }
    int32_t get_signed(int k) {
        if (k == 0) return 0;
        const uint32_t v = get(k);
        return k == 32 ? (int32_t)v : (int32_t)(v << (32 - k)) >> (32 - k);
    }
    uint32_t unary() {
        uint32_t q = 0;
        while (!bad && get(1) == 0) ++q;
        return q;
    }
    void align() { pos = (pos + 7) & ~(size_t)7; }
};

struct Md5 {  // RFC 1321
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
    uint64_t len = 0;
    uint8_t tail[64];
    size_t ntail = 0;
    static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
    void block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1,
            0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453,
            0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942,
            0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05,
            0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d,
            0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
        static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                  4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
        uint32_t m[16];
        for (int i = 0; i < 16; ++i) m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        uint32_t A = a, B = b, C = c, D = d;
#define ACE_MD5_STEP(F, G, i)                                   \
    {                                                           \
        const uint32_t f_ = (F), t_ = D;                        \
        D = C, C = B, B = B + rol(A + f_ + K[i] + m[(G)&15], S[i]), A = t_; \
    }
#pragma unroll
        for (int i = 0; i < 16; ++i) ACE_MD5_STEP((B & C) | (~B & D), i, i)
#pragma unroll
        for (int i = 16; i < 32; ++i) ACE_MD5_STEP((D & B) | (~D & C), 5 * i + 1, i)
#pragma unroll
        for (int i = 32; i < 48; ++i) ACE_MD5_STEP(B ^ C ^ D, 3 * i + 5, i)
#pragma unroll
        for (int i = 48; i < 64; ++i) ACE_MD5_STEP(C ^ (B | ~D), 7 * i, i)
#undef ACE_MD5_STEP
        a += A, b += B, c += C, d += D;
    }
    void update(const uint8_t* p, size_t n) {
        len += n;
        if (ntail) {
            const size_t take = n < 64 - ntail ? n : 64 - ntail;
            memcpy(tail + ntail, p, take);
            ntail += take, p += take, n -= take;
            if (ntail == 64) {
                block(tail);
                ntail = 0;
            }
        }
        for (; n >= 64; p += 64, n -= 64) block(p);
        if (n) {
            memcpy(tail, p, n);
            ntail = n;
        }
    }
    void finish(uint8_t out[16]) {
        const uint64_t bits = len * 8;
        const uint8_t one = 0x80, zero = 0;
        update(&one, 1);
        while (ntail != 56) update(&zero, 1);
        uint8_t lb[8];
        for (int i = 0; i < 8; ++i) lb[i] = (uint8_t)(bits >> (8 * i));
        update(lb, 8);
        const uint32_t w[4] = {a, b, c, d};
        for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(w[i / 4] >> (8 * (i % 4)));
    }
};

// ------------------------------------------------------------------------------------------------ FLAC encoder
struct SubPlan {
    int type = 1;  // 0 constant, 1 verbatim, 2 fixed
    int order = 0, porder = 0;
    uint8_t k[256];
    uint64_t bits = 0;
};

inline uint32_t fold(int32_t r) { return ((uint32_t)r << 1) ^ (uint32_t)(r >> 31); }

void fixed_residual(const int32_t* s, int n, int order, int32_t* r) {
    switch (order) {
        case 0: for (int i = 0; i < n; ++i) r[i] = s[i]; break;
        case 1: for (int i = 1; i < n; ++i) r[i] = s[i] - s[i - 1]; break;
        case 2: for (int i = 2; i < n; ++i) r[i] = s[i] - 2 * s[i - 1] + s[i - 2]; break;
        case 3: for (int i = 3; i < n; ++i) r[i] = s[i] - 3 * s[i - 1] + 3 * s[i - 2] - s[i - 3]; break;
        default: for (int i = 4; i < n; ++i) r[i] = s[i] - 4 * s[i - 1] + 6 * s[i - 2] - 4 * s[i - 3] + s[i - 4]; break;
    }
}

// Rice parameter of one partition from its sample count and the sum of the folded residuals; returns the estimated size.
inline uint64_t best_rice(uint64_t n, uint64_t sum, uint8_t* k_out) {
    uint64_t best = ~0ull;
    int bk = 0;
    for (int k = 0; k <= 14; ++k) {
        const uint64_t b = n * (uint64_t)(k + 1) + (sum >> k);
        if (b < best) best = b, bk = k;
    }
    *k_out = (uint8_t)bk;
    return best + 4;
}

// Plans one subframe of `n` samples of `bps` bits and leaves the residual of the chosen order in `res`.
void plan_subframe(const int32_t* s, int n, int bps, int32_t* res, SubPlan* pl) {
    bool constant = true;
    for (int i = 1; i < n && constant; ++i) constant = s[i] == s[0];
    if (constant) {
        pl->type = 0;
        pl->bits = 8 + (uint64_t)bps;
        return;
    }
    const uint64_t verbatim_bits = 8 + (uint64_t)n * bps;
    pl->type = 1;
    pl->bits = verbatim_bits;
    if (n <= 4) return;
    // sum |residual| of the five fixed predictors in one pass (running differences)
    uint64_t tot[5] = {0, 0, 0, 0, 0};
    int32_t l0 = s[3], l1 = s[3] - s[2], l2 = l1 - (s[2] - s[1]), l3 = l2 - ((s[2] - s[1]) - (s[1] - s[0]));
    for (int i = 4; i < n; ++i) {
        const int32_t d0 = s[i], d1 = d0 - l0, d2 = d1 - l1, d3 = d2 - l2, d4 = d3 - l3;
        tot[0] += (uint32_t)(d0 < 0 ? -d0 : d0), tot[1] += (uint32_t)(d1 < 0 ? -d1 : d1), tot[2] += (uint32_t)(d2 < 0 ? -d2 : d2);
        tot[3] += (uint32_t)(d3 < 0 ? -d3 : d3), tot[4] += (uint32_t)(d4 < 0 ? -d4 : d4);
        l0 = d0, l1 = d1, l2 = d2, l3 = d3;
    }
    int order = 0;
    for (int o = 1; o <= 4; ++o)
        if (tot[o] < tot[order]) order = o;
    fixed_residual(s, n, order, res);
    int maxp = 0;
    while (maxp < 8 && ((n >> (maxp + 1)) << (maxp + 1)) == n && (n >> (maxp + 1)) > order) ++maxp;
    uint64_t sums[256];
    const int np = 1 << maxp, plen = n >> maxp;
    for (int p = 0; p < np; ++p) {
        uint64_t a = 0;
        for (int i = (p == 0 ? order : p * plen); i < (p + 1) * plen; ++i) a += fold(res[i]);
        sums[p] = a;
    }
    uint64_t best = ~0ull;
    uint8_t ks[256];
    for (int po = maxp; po >= 0; --po) {
        const int cnt = 1 << po, len = n >> po;
        uint64_t bits = 0;
        for (int p = 0; p < cnt; ++p) bits += best_rice((uint64_t)(len - (p == 0 ? order : 0)), sums[p], ks + p);
        if (bits < best) {
            best = bits;
            pl->porder = po;
            memcpy(pl->k, ks, (size_t)cnt);
        }
        for (int p = 0; p < cnt / 2; ++p) sums[p] = sums[2 * p] + sums[2 * p + 1];
    }
    const uint64_t fixed_bits = 8 + (uint64_t)order * bps + 6 + best;
    if (fixed_bits < verbatim_bits) {
        pl->type = 2;
        pl->order = order;
        pl->bits = fixed_bits;
    }
}

void write_subframe(BitWriter& bw, const int32_t* s, int n, int bps, const int32_t* res, const SubPlan& pl) {
    if (pl.type == 0) {
        bw.put(0x00, 8);
        bw.put_signed(s[0], bps);
    } else if (pl.type == 1) {
        bw.put(0x02, 8);
        for (int i = 0; i < n; ++i) bw.put_signed(s[i], bps);
    } else {
        bw.put((uint32_t)((0x08 | pl.order) << 1), 8);
        for (int i = 0; i < pl.order; ++i) bw.put_signed(s[i], bps);
        bw.put(0, 2);  // Rice coding with 4-bit parameters
        bw.put((uint32_t)pl.porder, 4);
        const int cnt = 1 << pl.porder, len = n >> pl.porder;
        for (int p = 0; p < cnt; ++p) {
            const int k = pl.k[p];
            bw.put((uint32_t)k, 4);
            for (int i = (p == 0 ? pl.order : p * len); i < (p + 1) * len; ++i) {
                const uint32_t u = fold(res[i]);
                bw.unary(u >> k);
                bw.put(u, k);  // put() masks to the low k bits
            }
        }
    }
}

int sample_rate_code(int sr, int* extra_bits, uint32_t* extra) {
    static const int table[][2] = {{88200, 1}, {176400, 2}, {192000, 3}, {8000, 4}, {16000, 5}, {22050, 6}, {24000, 7}, {32000, 8}, {44100, 9}, {48000, 10}, {96000, 11}};
    *extra_bits = 0;
    for (auto& t : table)
        if (t[0] == sr) return t[1];
    if (sr % 1000 == 0 && sr / 1000 < 256) {
        *extra_bits = 8, *extra = (uint32_t)(sr / 1000);
        return 12;
    }
    if (sr < 65536) {
        *extra_bits = 16, *extra = (uint32_t)sr;
        return 13;
    }
    if (sr % 10 == 0 && sr / 10 < 65536) {
        *extra_bits = 16, *extra = (uint32_t)(sr / 10);
        return 14;
    }
    return 0;  // "see STREAMINFO"
}

// One frame: pcm = interleaved int16 at the first sample of the block.
// Per-thread scratch, allocated once per worker instead of ~150 KB per frame.
struct FrameScratch {
    // worst case: verbatim subframes (16 + 17 bits per stereo sample pair); the Rice size estimate is an upper bound of
    // the coded size (sum of floors <= floor of the sum), so a planned subframe never outgrows its verbatim form
    std::vector<uint8_t> bytes = std::vector<uint8_t>((size_t)kBlock * 2 * 17 / 8 + 64);
    std::vector<int32_t> sig = std::vector<int32_t>((size_t)4 * kBlock), res = std::vector<int32_t>((size_t)4 * kBlock);
};
void encode_frame(const int16_t* pcm, int n, int channels, int sample_rate, uint64_t frame_no, FrameScratch& sc, std::vector<uint8_t>& out) {
    std::vector<uint8_t>& bytes = sc.bytes;
    std::vector<int32_t>&sig = sc.sig, &res = sc.res;
    int32_t *L = sig.data(), *R = L + n, *M = R + n, *S = M + n;
    SubPlan pl[4];
    int assign = 0;  // channel assignment code
    const int32_t* sub[2];
    const int32_t* subres[2];
    const SubPlan* subpl[2];
    int subbps[2] = {16, 16};
    if (channels == 1) {
        for (int i = 0; i < n; ++i) L[i] = pcm[i];
        plan_subframe(L, n, 16, res.data(), &pl[0]);
        sub[0] = L, subres[0] = res.data(), subpl[0] = &pl[0];
    } else {
        for (int i = 0; i < n; ++i) {
            const int32_t l = pcm[2 * i], r = pcm[2 * i + 1];
            L[i] = l, R[i] = r, M[i] = (l + r) >> 1, S[i] = l - r;
        }
        for (int c = 0; c < 4; ++c) plan_subframe(sig.data() + (size_t)c * n, n, c == 3 ? 17 : 16, res.data() + (size_t)c * n, &pl[c]);
        const uint64_t b_lr = pl[0].bits + pl[1].bits, b_ls = pl[0].bits + pl[3].bits, b_sr = pl[3].bits + pl[1].bits, b_ms = pl[2].bits + pl[3].bits;
        int a = 0, b = 1;
        assign = 1;
        uint64_t best = b_lr;
        if (b_ls < best) best = b_ls, assign = 8, a = 0, b = 3;
        if (b_sr < best) best = b_sr, assign = 9, a = 3, b = 1;
        if (b_ms < best) best = b_ms, assign = 10, a = 2, b = 3;
        sub[0] = sig.data() + (size_t)a * n, sub[1] = sig.data() + (size_t)b * n;
        subres[0] = res.data() + (size_t)a * n, subres[1] = res.data() + (size_t)b * n;
        subpl[0] = &pl[a], subpl[1] = &pl[b];
        subbps[0] = a == 3 ? 17 : 16, subbps[1] = b == 3 ? 17 : 16;
    }
    BitWriter bw(bytes.data());
    int sr_bits = 0;
    uint32_t sr_extra = 0;
    const int sr_code = sample_rate_code(sample_rate, &sr_bits, &sr_extra);
    const int bs_code = n == kBlock ? 12 : (n <= 256 ? 6 : 7);
    bw.put(0xFFF8, 16);  // sync, reserved 0, fixed block size
    bw.put((uint32_t)((bs_code << 4) | sr_code), 8);
    bw.put((uint32_t)((assign << 4) | (4 << 1)), 8);  // 16 bits per sample
    // "UTF-8" coded frame number
    if (frame_no < 0x80) {
        bw.put((uint32_t)frame_no, 8);
    } else {
        int nb = 2;
        while (nb < 7 && (frame_no >> (5 * nb + 1)) != 0) ++nb;  // nb bytes carry 5*nb+1 payload bits (nb >= 2)
        bw.put((uint32_t)(((0xFF00u >> nb) & 0xFF) | (uint32_t)(frame_no >> (6 * (nb - 1)))), 8);
        for (int i = nb - 2; i >= 0; --i) bw.put((uint32_t)(0x80 | ((frame_no >> (6 * i)) & 0x3F)), 8);
    }
    if (bs_code == 6) bw.put((uint32_t)(n - 1), 8);
    if (bs_code == 7) bw.put((uint32_t)(n - 1), 16);
    if (sr_bits) bw.put(sr_extra, sr_bits);
    bw.put(crc().crc8(bytes.data(), bw.flush()), 8);
    for (int c = 0; c < channels; ++c) write_subframe(bw, sub[c], n, subbps[c], subres[c], *subpl[c]);
    bw.align();
    bw.put(crc().crc16(bytes.data(), bw.flush()), 16);
    out.assign(bytes.data(), bytes.data() + bw.flush());
}

int pick_threads(int n_threads, long jobs) {
    if (n_threads <= 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        n_threads = hc == 0 ? 4 : (hc > 16 ? 16 : (int)hc);
    }
    if ((long)n_threads > jobs) n_threads = (int)(jobs < 1 ? 1 : jobs);
    return n_threads;
}

template <class F>
void parallel_jobs(long jobs, int n_threads, F&& fn) {
    n_threads = pick_threads(n_threads, jobs);
    if (n_threads <= 1) {
        for (long j = 0; j < jobs; ++j) fn(j);
        return;
    }
    std::atomic<long> next{0};
    std::vector<std::thread> th;
    th.reserve(n_threads);
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&] {
            for (long j; (j = next.fetch_add(1)) < jobs;) fn(j);
        });
    for (auto& t : th) t.join();
}

void put_streaminfo(std::vector<uint8_t>& out, int64_t frames, int channels, int sample_rate, uint32_t min_fs, uint32_t max_fs, const uint8_t md5[16]) {
    uint8_t hdr[42];
    BitWriter bw(hdr);
    bw.put(0x664C6143u, 32);  // "fLaC"
    bw.put(0x80, 8);          // last metadata block, type 0 (STREAMINFO)
    bw.put(34, 24);
    bw.put(kBlock, 16);
    bw.put(kBlock, 16);
    bw.put(min_fs, 24);
    bw.put(max_fs, 24);
    bw.put((uint32_t)sample_rate, 20);
    bw.put((uint32_t)(channels - 1), 3);
    bw.put(15, 5);  // bits per sample - 1
    bw.put((uint32_t)((uint64_t)frames >> 32) & 0xF, 4);
    bw.put((uint32_t)((uint64_t)frames & 0xFFFFFFFFu), 32);
    for (int i = 0; i < 16; ++i) bw.put(md5[i], 8);
    bw.flush();
    out.insert(out.end(), hdr, hdr + 42);
}

// All (item, block) jobs of a batch share one pool; items[i] = interleaved PCM of `frames` frames.
void flac_encode_batch(const int16_t* const* items, int n_items, int64_t frames, int channels, int sample_rate, int n_threads,
                       std::vector<std::vector<uint8_t>>& files) {
    const long nblk = (long)((frames + kBlock - 1) / kBlock);
    std::vector<std::vector<uint8_t>> blocks((size_t)n_items * nblk);
    std::vector<uint8_t> md5((size_t)n_items * 16);
    // jobs [0, n_items) = MD5 of an item (long, scheduled first), the rest = frames
    parallel_jobs((long)n_items + (long)n_items * nblk, n_threads, [&](long j) {
        if (j < n_items) {
            Md5 m;
            m.update(reinterpret_cast<const uint8_t*>(items[j]), (size_t)frames * channels * 2);  // little-endian host
            m.finish(md5.data() + 16 * j);
            return;
        }
        j -= n_items;
        const long it = j / nblk, blk = j % nblk;
        const int64_t first = (int64_t)blk * kBlock;
        const int n = (int)(frames - first < kBlock ? frames - first : kBlock);
        static thread_local FrameScratch scratch;
        encode_frame(items[it] + first * channels, n, channels, sample_rate, (uint64_t)blk, scratch, blocks[(size_t)j]);
    });
    files.assign(n_items, {});
    for (int it = 0; it < n_items; ++it) {
        size_t total = 42;
        uint32_t mn = 0xFFFFFF, mx = 0;
        for (long b = 0; b < nblk; ++b) {
            const size_t sz = blocks[(size_t)it * nblk + b].size();
            total += sz;
            mn = sz < mn ? (uint32_t)sz : mn, mx = sz > mx ? (uint32_t)sz : mx;
        }
        if (nblk == 0) mn = 0;
        auto& f = files[it];
        f.reserve(total);
        put_streaminfo(f, frames, channels, sample_rate, mn, mx, md5.data() + 16 * it);
        for (long b = 0; b < nblk; ++b) {
            auto& blk = blocks[(size_t)it * nblk + b];
            f.insert(f.end(), blk.begin(), blk.end());
            std::vector<uint8_t>().swap(blk);
        }
    }
}

// ------------------------------------------------------------------------------------------------ FLAC decoder (16-bit)
// Used by convert_audio (audio_utils.py:217-257: load a file, save it in another format) and by the full-size round-trip
// tests.  Subset: <= 16 bits per sample, 1-2 channels; constant / verbatim / fixed / LPC subframes, Rice and Rice2.
const char* decode_subframe(BitReader& br, int n, int bps, int32_t* s) {
    if (br.get(1)) return "subframe padding bit set";
    const int type = (int)br.get(6);
    int wasted = 0;
    if (br.get(1)) wasted = 1 + (int)br.unary();
    bps -= wasted;
    if (bps <= 0 || bps > 32) return "bad wasted bits";
    int order = 0;
    if (type == 0) {
        const int32_t v = br.get_signed(bps);
        for (int i = 0; i < n; ++i) s[i] = v;
    } else if (type == 1) {
        for (int i = 0; i < n; ++i) s[i] = br.get_signed(bps);
    } else if ((type >= 8 && type <= 12) || type >= 32) {
        const bool lpc = type >= 32;
        order = lpc ? (type & 31) + 1 : (type & 7);
        if (order > n) return "predictor order exceeds the block";
        for (int i = 0; i < order; ++i) s[i] = br.get_signed(bps);
        int32_t coef[32];
        int shift = 0;
        if (lpc) {
            const int prec = (int)br.get(4) + 1;
            if (prec == 16) return "bad LPC precision";
            shift = br.get_signed(5);
            if (shift < 0) return "negative LPC shift";
            for (int i = 0; i < order; ++i) coef[i] = br.get_signed(prec);
        }
        const int method = (int)br.get(2);
        if (method > 1) return "reserved residual coding method";
        const int kbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
        const int po = (int)br.get(4), cnt = 1 << po;
        if ((n >> po) << po != n && po > 0) return "partition order does not divide the block";
        const int len = n >> po;
        if (len < order && po > 0) return "partition shorter than the predictor";
        int i = order;
        for (int p = 0; p < cnt; ++p) {
            const int k = (int)br.get(kbits);
            const int end = (p + 1) * len;
            if (k == esc) {
                const int raw = (int)br.get(5);
                for (; i < end; ++i) s[i] = br.get_signed(raw);
            } else {
                for (; i < end; ++i) {
                    const uint32_t q = br.unary();
                    const uint32_t u = (q << k) | br.get(k);
                    s[i] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
                }
            }
            if (br.bad) return "truncated residual";
        }
        if (lpc) {
            for (int j = order; j < n; ++j) {
                int64_t acc = 0;
                for (int c = 0; c < order; ++c) acc += (int64_t)coef[c] * s[j - 1 - c];
                s[j] += (int32_t)(acc >> shift);
            }
        } else {
            switch (order) {
                case 0: break;
                case 1: for (int j = 1; j < n; ++j) s[j] += s[j - 1]; break;
                case 2: for (int j = 2; j < n; ++j) s[j] += 2 * s[j - 1] - s[j - 2]; break;
                case 3: for (int j = 3; j < n; ++j) s[j] += 3 * s[j - 1] - 3 * s[j - 2] + s[j - 3]; break;
                default: for (int j = 4; j < n; ++j) s[j] += 4 * s[j - 1] - 6 * s[j - 2] + 4 * s[j - 3] - s[j - 4]; break;
            }
        }
    } else {
        return "reserved subframe type";
    }
    if (wasted)
        for (int i = 0; i < n; ++i) s[i] = (int32_t)((uint32_t)s[i] << wasted);
    return br.bad ? "truncated subframe" : nullptr;
}

struct FlacInfo {
    int64_t frames = 0;
    int channels = 0, sample_rate = 0, bps = 0;
    size_t first_frame = 0;
    uint8_t md5[16];
};

const char* flac_parse_header(const uint8_t* p, size_t n, FlacInfo* fi) {
    if (n < 42 || memcmp(p, "fLaC", 4) != 0) return "not a FLAC stream";
    size_t pos = 4;
    bool have = false;
    for (;;) {
        if (pos + 4 > n) return "truncated metadata";
        const bool last = p[pos] & 0x80;
        const int type = p[pos] & 0x7F;
        const size_t len = ((size_t)p[pos + 1] << 16) | ((size_t)p[pos + 2] << 8) | p[pos + 3];
        pos += 4;
        if (pos + len > n) return "truncated metadata";
        if (type == 0) {
            if (len < 34) return "short STREAMINFO";
            BitReader br(p + pos, len);
            br.get(16), br.get(16), br.get(24), br.get(24);
            fi->sample_rate = (int)br.get(20);
            fi->channels = (int)br.get(3) + 1;
            fi->bps = (int)br.get(5) + 1;
            fi->frames = ((int64_t)br.get(4) << 32) | br.get(32);
            for (int i = 0; i < 16; ++i) fi->md5[i] = (uint8_t)br.get(8);
            have = true;
        }
        pos += len;
        if (last) break;
    }
    if (!have) return "no STREAMINFO";
    fi->first_frame = pos;
    return nullptr;
}

const char* flac_decode(const uint8_t* p, size_t n, const FlacInfo& fi, int16_t* out) {
    if (fi.bps > 16 || fi.channels < 1 || fi.channels > 2) return "decoder subset: <= 16 bits, 1-2 channels";
    size_t pos = fi.first_frame;
    int64_t done = 0;
    std::vector<int32_t> buf((size_t)2 * 65536);
    while (done < fi.frames) {
        if (pos + 6 > n) return "truncated stream";
        if (p[pos] != 0xFF || (p[pos + 1] & 0xFE) != 0xF8) return "lost frame sync";
        const bool variable = p[pos + 1] & 1;
        (void)variable;
        BitReader br(p + pos, n - pos);
        br.get(16);
        const int bs_code = (int)br.get(4), sr_code = (int)br.get(4), assign = (int)br.get(4), ss_code = (int)br.get(3);
        if (br.get(1)) return "reserved header bit set";
        uint32_t first = br.get(8);  // UTF-8 coded frame / sample number: skip the continuation bytes
        int extra = 0;
        while (first & 0x80) {
            first <<= 1;
            ++extra;
        }
        for (int i = 1; i < extra; ++i) br.get(8);
        int bs;
        if (bs_code == 0) return "reserved block size";
        else if (bs_code == 1) bs = 192;
        else if (bs_code <= 5) bs = 576 << (bs_code - 2);
        else if (bs_code == 6) bs = (int)br.get(8) + 1;
        else if (bs_code == 7) bs = (int)br.get(16) + 1;
        else bs = 256 << (bs_code - 8);
        if (sr_code == 12) br.get(8);
        else if (sr_code == 13 || sr_code == 14) br.get(16);
        else if (sr_code == 15) return "invalid sample rate code";
        static const int ss_tab[8] = {0, 8, 12, -1, 16, 20, 24, 32};
        const int bps = ss_code == 0 ? fi.bps : ss_tab[ss_code];
        if (bps <= 0 || bps > 16) return "unsupported sample size";
        const size_t hdr = br.pos >> 3;
        if (crc().crc8(p + pos, hdr) != br.get(8)) return "frame header CRC-8 mismatch";
        const int nch = assign < 8 ? assign + 1 : 2;
        if (assign > 10 || nch != fi.channels) return "channel assignment does not match STREAMINFO";
        if (bs > 65536 || done + bs > fi.frames) return "block overruns the stream";
        int32_t *c0 = buf.data(), *c1 = c0 + 65536;
        for (int c = 0; c < nch; ++c) {
            const bool side = (assign == 8 && c == 1) || (assign == 9 && c == 0) || (assign == 10 && c == 1);
            if (const char* e = decode_subframe(br, bs, bps + (side ? 1 : 0), c == 0 ? c0 : c1)) return e;
        }
        br.align();
        const size_t body = br.pos >> 3;
        if (crc().crc16(p + pos, body) != br.get(16) || br.bad) return "frame CRC-16 mismatch";
        int16_t* o = out + done * nch;
        if (nch == 1) {
            for (int i = 0; i < bs; ++i) o[i] = (int16_t)c0[i];
        } else {
            for (int i = 0; i < bs; ++i) {
                int32_t l, r;
                if (assign == 8) l = c0[i], r = c0[i] - c1[i];
                else if (assign == 9) l = c0[i] + c1[i], r = c1[i];
                else if (assign == 10) {
                    const int32_t s = c1[i], m = (int32_t)(((uint32_t)c0[i] << 1) | (uint32_t)(s & 1));
                    l = (m + s) >> 1, r = (m - s) >> 1;
                } else l = c0[i], r = c1[i];
                o[2 * i] = (int16_t)l, o[2 * i + 1] = (int16_t)r;
            }
        }
        done += bs;
        pos += br.pos >> 3;
    }
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ RIFF/WAVE
void put_le(std::vector<uint8_t>& o, uint32_t v, int bytes) {
    for (int i = 0; i < bytes; ++i) o.push_back((uint8_t)(v >> (8 * i)));
}
// is_float: IEEE float32 (what torchaudio's soundfile backend writes for a float32 tensor, "wav" and "wav32" alike) with
// the `fact` chunk the format asks of non-PCM data; otherwise PCM_16.
void wav_header(std::vector<uint8_t>& o, int64_t frames, int channels, int sample_rate, bool is_float) {
    const uint32_t bytes_per = is_float ? 4 : 2, data = (uint32_t)(frames * channels * bytes_per);
    const uint32_t riff = 4 + (8 + 16) + (is_float ? 12 : 0) + 8 + data;
    o.insert(o.end(), {'R', 'I', 'F', 'F'});
    put_le(o, riff, 4);
    o.insert(o.end(), {'W', 'A', 'V', 'E', 'f', 'm', 't', ' '});
    put_le(o, 16, 4);
    put_le(o, is_float ? 3 : 1, 2);
    put_le(o, (uint32_t)channels, 2);
    put_le(o, (uint32_t)sample_rate, 4);
    put_le(o, (uint32_t)sample_rate * channels * bytes_per, 4);
    put_le(o, channels * bytes_per, 2);
    put_le(o, 8 * bytes_per, 2);
    if (is_float) {
        o.insert(o.end(), {'f', 'a', 'c', 't'});
        put_le(o, 4, 4);
        put_le(o, (uint32_t)frames, 4);
    }
    o.insert(o.end(), {'d', 'a', 't', 'a'});
    put_le(o, data, 4);
}

int write_file(const char* path, const uint8_t* a, size_t na, const uint8_t* b, size_t nb) {
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    const bool ok = (na == 0 || fwrite(a, 1, na, f) == na) && (nb == 0 || fwrite(b, 1, nb, f) == nb);
    return (fclose(f) == 0 && ok) ? 0 : 1;
}

}  // namespace
}  // namespace ace355

using namespace ace355;

extern "C" {

int ace355_normalize_audio(float* wav_dev, int n_items, int64_t per_item, float target_db, float* peaks_host, void* stream) {
    ACE_CHECK(wav_dev && n_items > 0 && n_items <= 4096 && per_item > 0, "normalize_audio: bad argument");
    float* peaks = nullptr;
    ACE_HIP(hipMalloc((void**)&peaks, sizeof(float) * n_items));
    // 10 ** (target_db / 20.0) is evaluated in double by the reference and rounded to fp32 when it meets the tensor
    const float amp = (float)pow(10.0, (double)target_db / 20.0);
    int rc = launch_normalize_db(wav_dev, n_items, (long)per_item, amp, peaks, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (rc == 0 && e == hipSuccess && peaks_host) e = hipMemcpy(peaks_host, peaks, sizeof(float) * n_items, hipMemcpyDeviceToHost);
    hipFree(peaks);
    if (e != hipSuccess) return hip_fail(e, "normalize_audio", __FILE__, __LINE__);
    return rc;
}

int ace355_audio_interleave(const float* wav_dev, int n_items, int channels, int64_t samples, void* out_dev, int as_pcm16, void* stream) {
    ACE_CHECK(wav_dev && out_dev && n_items > 0 && samples > 0 && (channels == 1 || channels == 2), "audio_interleave: bad argument");
    return launch_interleave(wav_dev, n_items, channels, (long)samples, out_dev, as_pcm16, (hipStream_t)stream);
}

int64_t ace355_flac_bound(int64_t frames, int channels) {
    if (frames < 0 || channels < 1) return -1;
    const int64_t nblk = (frames + kBlock - 1) / kBlock;
    return 42 + nblk * 32 + (frames * channels * 17 + 7) / 8 + nblk * channels * 2;  // verbatim worst case (side = 17 bits)
}

int ace355_flac_encode_pcm16(const int16_t* pcm, int64_t frames, int channels, int sample_rate, int n_threads, uint8_t* out, int64_t cap,
                             int64_t* out_len) {
    ACE_CHECK(out && out_len && frames >= 0 && (frames == 0 || pcm) && (channels == 1 || channels == 2), "flac_encode_pcm16: bad argument");
    ACE_CHECK(sample_rate > 0 && sample_rate < (1 << 20) && frames < (1ll << 36), "flac_encode_pcm16: sample rate / length outside STREAMINFO");
    std::vector<std::vector<uint8_t>> files;
    const int16_t* items[1] = {pcm};
    flac_encode_batch(items, 1, frames, channels, sample_rate, n_threads, files);
    *out_len = (int64_t)files[0].size();
    ACE_CHECK(*out_len <= cap, "flac_encode_pcm16: output buffer too small (see ace355_flac_bound)");
    memcpy(out, files[0].data(), files[0].size());
    return ACE355_OK;
}

int ace355_flac_info(const uint8_t* data, int64_t size, int64_t* frames, int32_t* channels, int32_t* sample_rate, int32_t* bits_per_sample) {
    ACE_CHECK(data && size > 0, "flac_info: bad argument");
    FlacInfo fi;
    if (const char* e = flac_parse_header(data, (size_t)size, &fi)) {
        set_error(std::string("flac_info: ") + e);
        return ACE355_ERR_INVALID;
    }
    if (frames) *frames = fi.frames;
    if (channels) *channels = fi.channels;
    if (sample_rate) *sample_rate = fi.sample_rate;
    if (bits_per_sample) *bits_per_sample = fi.bps;
    return ACE355_OK;
}

int ace355_flac_decode_pcm16(const uint8_t* data, int64_t size, int16_t* pcm_out, int64_t cap_samples, int verify_md5) {
    ACE_CHECK(data && size > 0 && pcm_out, "flac_decode_pcm16: bad argument");
    FlacInfo fi;
    const char* e = flac_parse_header(data, (size_t)size, &fi);
    if (!e && fi.frames * fi.channels > cap_samples) e = "output buffer too small";
    if (!e) e = flac_decode(data, (size_t)size, fi, pcm_out);
    if (!e && verify_md5) {
        uint8_t zero[16] = {0}, got[16];
        if (memcmp(fi.md5, zero, 16) != 0) {
            Md5 m;
            m.update(reinterpret_cast<const uint8_t*>(pcm_out), (size_t)fi.frames * fi.channels * 2);
            m.finish(got);
            if (memcmp(got, fi.md5, 16) != 0) e = "MD5 of the decoded PCM does not match STREAMINFO";
        }
    }
    if (e) {
        set_error(std::string("flac_decode_pcm16: ") + e);
        return ACE355_ERR_INVALID;
    }
    return ACE355_OK;
}

int64_t ace355_wav_bound(int64_t frames, int channels, int is_float) { return 64 + frames * channels * (is_float ? 4 : 2); }

int ace355_wav_encode(const void* interleaved, int64_t frames, int channels, int sample_rate, int is_float, uint8_t* out, int64_t cap,
                      int64_t* out_len) {
    ACE_CHECK(out && out_len && frames >= 0 && (frames == 0 || interleaved) && channels >= 1 && channels <= 8 && sample_rate > 0,
              "wav_encode: bad argument");
    const int64_t bytes = frames * channels * (is_float ? 4 : 2);
    ACE_CHECK(bytes < 0xFFFFFF00ll - 64, "wav_encode: RIFF holds < 4 GiB");
    std::vector<uint8_t> h;
    wav_header(h, frames, channels, sample_rate, is_float != 0);
    *out_len = (int64_t)h.size() + bytes;
    ACE_CHECK(*out_len <= cap, "wav_encode: output buffer too small (see ace355_wav_bound)");
    memcpy(out, h.data(), h.size());
    if (bytes) memcpy(out + h.size(), interleaved, (size_t)bytes);
    return ACE355_OK;
}

int ace355_save_audio_batch(const float* wav_dev, int n_items, int channels, int64_t samples, int sample_rate, int format,
                            const char* const* paths, int n_threads, void* stream) {
    ACE_CHECK(wav_dev && paths && n_items > 0 && n_items <= 4096 && samples > 0 && (channels == 1 || channels == 2), "save_audio_batch: bad argument");
    ACE_CHECK(format >= ACE355_AUDIO_FLAC && format <= ACE355_AUDIO_WAV_PCM16, "save_audio_batch: format is FLAC (PCM_16), WAV float32 or WAV PCM_16");
    ACE_CHECK(sample_rate > 0 && sample_rate < (1 << 20), "save_audio_batch: bad sample rate");
    for (int i = 0; i < n_items; ++i) ACE_CHECK(paths[i] && paths[i][0], "save_audio_batch: empty path");
    const bool is_float = format == ACE355_AUDIO_WAV_F32;
    const size_t per_item = (size_t)samples * channels * (is_float ? 4 : 2), total = per_item * n_items;
    // conversion scratch (device) and its pinned host mirror are kept between calls: page-locking 46 MB per batch cost
    // more than the copy.  One call at a time holds them.
    static std::mutex scratch_mutex;
    static void* s_dev = nullptr;
    static uint8_t* s_host = nullptr;
    static size_t s_cap = 0;
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (total > s_cap) {
        if (s_dev) hipFree(s_dev);
        if (s_host) hipHostFree(s_host);
        s_dev = nullptr, s_host = nullptr, s_cap = 0;
        ACE_HIP(hipMalloc(&s_dev, total));
        const hipError_t ea = hipHostMalloc((void**)&s_host, total, hipHostMallocDefault);
        if (ea != hipSuccess) {
            hipFree(s_dev);
            s_dev = nullptr;
            return hip_fail(ea, "save_audio_batch: pinned scratch", __FILE__, __LINE__);
        }
        s_cap = total;
    }
    void* dev = s_dev;
    uint8_t* host = s_host;
    int rc = launch_interleave(wav_dev, n_items, channels, (long)samples, dev, is_float ? 0 : 1, (hipStream_t)stream);
    hipError_t e = rc == 0 ? hipMemcpyAsync(host, dev, total, hipMemcpyDeviceToHost, (hipStream_t)stream) : hipSuccess;
    if (e == hipSuccess && rc == 0) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess || rc != 0) return e != hipSuccess ? hip_fail(e, "save_audio_batch", __FILE__, __LINE__) : rc;
    std::atomic<int> failed{-1};
    if (format == ACE355_AUDIO_FLAC) {
        std::vector<const int16_t*> items(n_items);
        for (int i = 0; i < n_items; ++i) items[i] = reinterpret_cast<const int16_t*>(host + per_item * i);
        std::vector<std::vector<uint8_t>> files;
        flac_encode_batch(items.data(), n_items, samples, channels, sample_rate, n_threads, files);
        parallel_jobs(n_items, n_threads, [&](long i) {
            if (write_file(paths[i], files[i].data(), files[i].size(), nullptr, 0)) failed = (int)i;
        });
    } else {
        std::vector<uint8_t> h;
        wav_header(h, samples, channels, sample_rate, is_float);
        parallel_jobs(n_items, n_threads, [&](long i) {
            if (write_file(paths[i], h.data(), h.size(), host + per_item * i, per_item)) failed = (int)i;
        });
    }
    if (failed >= 0) {
        set_error(std::string("save_audio_batch: cannot write ") + paths[failed.load()]);
        return ACE355_ERR_INVALID;
    }
    return ACE355_OK;
}

}  // extern "C"
