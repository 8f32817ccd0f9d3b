// attn.hip - bidirectional flash attention for the DiT (full / sliding-band self-attention, cross-attention).
//
// Replaces ALL_ATTENTION_FUNCTIONS[impl] / eager_attention_forward as called from AceStepAttention.forward
// (base.py:351-367) including the dense additive band mask of create_4d_mask (base.py:56-135): the band predicate
// |i-j| <= window is evaluated in-kernel and K/V tiles wholly outside the band are skipped.
//
// gfx950 design (head_dim 128, GQA): one workgroup = 128 query rows of one (sequence, q-head); 4 waves x 32 rows.
// K tiles [64 keys][128 d] and V^T tiles [128 d][64 keys] are staged in LDS (XOR-swizzled, conflict-free
// ds_read_b128 / ds_read_b64).  QK^T is computed "swapped" (S^T = K Q^T, MFMA 32x32x16) so a lane owns one query
// column: the online-softmax row max/sum are in-register (+1 cross-half shuffle), and the C-layout of P is
// already the B-operand layout needed by O^T += V^T P^T - no LDS round trip, no lane permutes for P.
// V is consumed pre-transposed ([d][key], produced by transpose_v_kernel / the cross-KV cache builder).
#include "common.h"

#include <stdio.h>
#include <stdlib.h>

namespace ace355 {

namespace {

constexpr int KB = 64;   // keys per tile

// ACE355_ATTN_CLK=1 (diagnostic): one wave of workgroup (0,0,0) records shader-clock totals: whole kernel, K loop, and the
// part of the loop spent in the per-tile wait + barrier; launch_attention prints them.
__device__ unsigned long long g_attn_probe[8];

__device__ __forceinline__ int k_off(int key, int slot) { return key * 256 + ((slot ^ (key & 15)) << 4); }
// V^T tile: row d = 128 B (64 keys); 8-byte chunk c8 (4 keys) stored at c8 ^ ((d>>1)&15)
__device__ __forceinline__ int vt_off8(int d, int c8) { return d * 128 + ((c8 ^ ((d >> 1) & 15)) << 3); }

// NW = waves per workgroup = 32-query row groups (4: 128 queries, 3: 96 queries - picked by the launcher so that the
// workgroup count is a whole number of rounds of the 512 resident slots).
template <int NW, int OCC = 2>  // OCC = workgroups per CU the register budget is sized for (3: <= 168 VGPRs)
__global__ __launch_bounds__(NW * 64, OCC) void attn_kernel(AttnArgs a, float scale_log2) {
    constexpr int QB = NW * 32;
    constexpr int NT = NW * 64;                       // threads
    constexpr int NI = (1024 + NT - 1) / NT;          // staging passes over the 1024 16-B chunks of a K (or V^T) tile
    __shared__ __attribute__((aligned(16))) char smem[32768];  // K tile 16 KB | V^T tile 16 KB
    char* Ks = smem;
    char* Vs = smem + 16384;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int hkv = h / (a.Hq / a.Hkv);
    const int q0 = qb * QB;
    const int lq = lane & 31, half = lane >> 5;
    const int qw0 = q0 + wave * 32;        // first query row of this wave
    const int qrow = qw0 + lq;             // this lane's query row
    const int qrow_c = min(qrow, a.Sq - 1);
    const int win = a.window < 0 ? (1 << 28) : a.window;  // "no band" as a band wider than any sequence

    // Q fragments (B operand of S^T = K Q^T): lane holds q = lq, d = ks*16 + half*8 .. +8
    bf16x8 qf[8];
    {
        const bf16_t* qp = a.q + (long)n * a.q_seq_stride + (long)qrow_c * a.q_row_stride + h * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = as_bf16x8(ldg16(qp + ks * 16));
    }

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;  // running max in log2-scaled units

    // key tile range for this query block
    int kt_lo = 0, kt_hi = (a.Skv + KB - 1) / KB;
    if (a.window >= 0) {
        kt_lo = max(0, q0 - a.window) / KB;
        kt_hi = min(kt_hi, (min(a.Skv - 1, q0 + QB - 1 + a.window)) / KB + 1);
    }

    const bf16_t* kbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.k_tab[n]) : a.k + (long)n * a.k_seq_stride) + (long)hkv * a.k_head_stride;
    const bf16_t* vbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.vt_tab[n]) : a.vt + (long)n * a.vt_seq_stride) + (long)hkv * a.vt_head_stride;

    // staging assignment: chunk c = tid + i*NT;  K: key = c>>4, slot = c&15;  V^T: d = c>>3, j = c&7
    uint4 rk[NI], rv[NI];
    auto load_kv = [&](int key0) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * NT;
            if (NI * NT == 1024 || c < 1024) {
                const int key = c >> 4, slot = c & 15;
                rk[i] = ldg16(kbase + (long)min(key0 + key, a.Skv - 1) * a.k_row_stride + slot * 8);
                const int d = c >> 3, j = c & 7;
                rv[i] = ldg16(vbase + (long)d * a.vt_ld + key0 + j * 8);
            }
        }
    };
    auto store_kv = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = tid + i * NT;
            if (NI * NT == 1024 || c < 1024) {
                const int key = c >> 4, slot = c & 15;
                *reinterpret_cast<uint4*>(Ks + k_off(key, slot)) = rk[i];
                const int d = c >> 3, j = c & 7;
                const int x = (d >> 1) & 15;
                uint4 v = rv[i];
                if (x & 1) v = make_uint4(v.z, v.w, v.x, v.y);  // the two 8-B halves swap places under the XOR
                *reinterpret_cast<uint4*>(Vs + d * 128 + ((j ^ (x >> 1)) << 4)) = v;
            }
        }
    };

    if (kt_lo < kt_hi) load_kv(kt_lo * KB);
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int key0 = kt * KB;
        __syncthreads();  // previous tile fully consumed
        store_kv();
        __syncthreads();
        if (kt + 1 < kt_hi) load_kv(key0 + KB);  // next tile's HBM/L2 latency hides under this tile's MFMAs

        // ---- S^T[key][q] = sum_d K[key][d] Q[q][d]
        f32x16 s[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 kf = as_bf16x8(*reinterpret_cast<const uint4*>(Ks + k_off(t2 * 32 + lq, ks * 2 + half)));
                s[t2] = mfma32(kf, qf[ks], s[t2]);
            }
        }

        // ---- mask (edge tiles only; wave-uniform test) + online softmax.  Lane owns query qrow; its 32 scores are
        //      keys key0 + t2*32 + (r&3) + 8*(r>>2) + 4*half.
        const bool interior = (key0 + KB <= a.Skv) && (qw0 + 31 - key0 <= win) && (key0 + KB - 1 - qw0 <= win);
        if (!interior) {
            const int kb = key0 + 4 * half;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + t2 * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = (key < a.Skv) & ((unsigned)(qrow - key + win) <= (unsigned)(2 * win));
                    s[t2][r] = ok ? s[t2][r] : -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);  // m_run = -inf -> 0
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t2][r], scale_log2, -m_use));
                s[t2][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {  // wave-uniform: skip the 64-multiply rescale when no row's max moved
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }

        // ---- O^T[d][q] += sum_key V^T[d][key] P^T[key][q]
        // k-slot e of MFMA step (t2, t) on lane-half `half` <-> key t2*32 + 16t + 4*half + (e&3) + 8*(e>>2):
        // exactly registers r = 8t .. 8t+7 of s[t2] (C layout), so P needs no data movement.
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint4 pb;
                pb.x = pack_bf2(s[t2][8 * t + 0], s[t2][8 * t + 1]);
                pb.y = pack_bf2(s[t2][8 * t + 2], s[t2][8 * t + 3]);
                pb.z = pack_bf2(s[t2][8 * t + 4], s[t2][8 * t + 5]);
                pb.w = pack_bf2(s[t2][8 * t + 6], s[t2][8 * t + 7]);
                const bf16x8 pf = as_bf16x8(pb);
                const int c8 = t2 * 8 + t * 4 + half;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int d = dt * 32 + lq;
                    const uint2 v0 = *reinterpret_cast<const uint2*>(Vs + vt_off8(d, c8));
                    const uint2 v1 = *reinterpret_cast<const uint2*>(Vs + vt_off8(d, c8 + 2));
                    const bf16x8 vf = as_bf16x8(make_uint4(v0.x, v0.y, v1.x, v1.y));
                    o[dt] = mfma32(vf, pf, o[dt]);
                }
            }
    }

    // ---- finalize: O[q][d] = O^T[d][q] / l ; lane holds d = dt*32 + 8g + 4*half + {0..3} for g = 0..3
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow < a.Sq) {
        bf16_t* op = a.out + (long)n * a.o_seq_stride + (long)qrow * a.o_row_stride + h * 128 + 4 * half;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint2 pk = make_uint2(pack_bf2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv),
                                            pack_bf2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv));
                *reinterpret_cast<uint2*>(op + dt * 32 + 8 * g) = pk;
            }
    }
}


// ------------------------------------------------------------------------------------------------ v3
// One workgroup = NW x 32 query rows of one (sequence, q-head), 2 waves per SIMD (256-VGPR budget; a 12-wave / 3-per-SIMD
// variant that covers Sq = 375 in one workgroup per CU was tried: at 168 VGPRs hipcc serialises every ds_read behind an
// lgkmcnt(0) or spills, and it lost to this one).  K / V^T tiles go HBM/L2 -> LDS by DMA (global_load_lds, asm-issued)
// into a double buffer: one barrier per tile, the next tile lands while this one is consumed, no staging registers; the
// 16 K fragments of a tile are read in ONE batch before the QK MFMAs, the 16 V^T fragments in one batch before the softmax.  Row permutation pi (swap bits 2 and 3 of the MFMA row index):
//   S^T = K Q^T reads K row pi(i) for MFMA row i, so that a lane's accumulator octet r = 8t..8t+7 is EIGHT CONTIGUOUS keys
//   (key = 8*half + 16*(r>>3) + (r&7)): the P^T operand of O^T += V^T P^T then pairs with ONE 16-byte V^T read, and both
//   tiles use the GEMM's 16-byte XOR swizzles (applied on the DMA source address).  The same permutation on the V^T rows
//   makes a lane's output octet eight contiguous d: the epilogue is 8 x 16-byte stores per lane.
__device__ __forceinline__ void attn_glds16(unsigned voff, const void* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ int pi23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn3_kernel(AttnArgs a, float scale_log2) {
    constexpr int QB = NW * 32;
    constexpr int BUF = 32768;                 // K tile 16 KB | V^T tile 16 KB
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const unsigned long long t_entry = clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    const int hkv = h / (a.Hq / a.Hkv);
    const int q0 = qb * QB;
    const int lq = lane & 31, half = lane >> 5;
    const int qw0 = q0 + wave * 32;
    const int qrow = qw0 + lq;
    const int qrow_c = min(qrow, a.Sq - 1);
    const int win = a.window < 0 ? (1 << 28) : a.window;
    const bool wave_live = qw0 < a.Sq;         // waves past the end of the sequence only help with the DMA
    const int skv = a.kv_len ? a.kv_len[n] : a.Skv;  // valid (un-padded) keys of this sequence

    int kt_lo = 0, kt_hi = (skv + KB - 1) / KB;
    if (a.window >= 0) {
        kt_lo = max(0, q0 - a.window) / KB;
        kt_hi = min(kt_hi, (min(skv - 1, q0 + QB - 1 + a.window)) / KB + 1);
    }
    const bf16_t* kbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.k_tab[n]) : a.k + (long)n * a.k_seq_stride) + (long)hkv * a.k_head_stride;
    const bf16_t* vbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.vt_tab[n]) : a.vt + (long)n * a.vt_seq_stride) + (long)hkv * a.vt_head_stride;

    // DMA piece p = wave + NW*i:  p < 16: K rows 4p..4p+3 (256 B each), lane = (row<<4 | slot');  p >= 16: V^T rows
    // 8(p-16)..+7 (128 B each), lane = (row<<3 | slot').  LDS image lane-linear; the XOR swizzle is on the source chunk.
    // Addressing = uniform 64-bit base in SGPRs (advances per tile) + a loop-invariant 32-bit per-lane byte offset: the
    // asm's VGPR operands are never recycled (hipcc waits vmcnt before it lets anything overwrite a register an inline-asm
    // VMEM instruction used, which would drain the prefetch in the middle of the QK MFMAs).
    static_assert(16 % NW == 0, "K and V^T pieces must not mix inside one i");
    constexpr int NPK = 16 / NW;  // K pieces per wave (i < NPK), V^T pieces per wave (i >= NPK)
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    unsigned k_voff[NPK], k_voff_tail[NPK], v_voff[NPK];
    const int tail_key0 = ((a.Skv - 1) / KB) * KB;  // first key of the last (possibly ragged) tile
#pragma unroll
    for (int i = 0; i < NPK; ++i) {
        const int p = wave + NW * i;
        const int key = 4 * p + (lane >> 4);
        const int j = (lane & 15) ^ (key & 15);
        k_voff[i] = (unsigned)(key * a.k_row_stride + j * 8) * 2u;
        k_voff_tail[i] = (unsigned)((min(tail_key0 + key, a.Skv - 1) - tail_key0) * a.k_row_stride + j * 8) * 2u;
        const int d = 8 * p + (lane >> 3);
        const int jv = (lane & 7) ^ ((d >> 1) & 7);
        v_voff[i] = (unsigned)(d * a.vt_ld + jv * 8) * 2u;
    }
    auto issue = [&](int kt) {
        const int key0 = kt * KB;
        const unsigned bb = lds0 + (unsigned)(kt & 1) * BUF;
        const bf16_t* kb_s = kbase + (long)key0 * a.k_row_stride;  // uniform
        const bf16_t* vb_s = vbase + key0;
        const bool tail = key0 + KB > a.Skv;
#pragma unroll
        for (int i = 0; i < NPK; ++i) {
            const unsigned pk = (unsigned)(wave + NW * i);
            if (tail) attn_glds16(k_voff_tail[i], kb_s, bb + pk * 1024u);
            else attn_glds16(k_voff[i], kb_s, bb + pk * 1024u);
        }
#pragma unroll
        for (int i = 0; i < NPK; ++i) {
            const unsigned pk = (unsigned)(wave + NW * i);
            attn_glds16(v_voff[i], vb_s, bb + 16384u + pk * 1024u);
        }
    };

    if (kt_lo < kt_hi) issue(kt_lo);

    // Q fragments (B operand of S^T = K Q^T): lane holds q = lq, d = ks*16 + half*8 .. +8
    bf16x8 qf[8];
    {
        const bf16_t* qp = a.q + (long)n * a.q_seq_stride + (long)qrow_c * a.q_row_stride + h * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = as_bf16x8(ldg16(qp + ks * 16));
    }
    // vmcnt(0) in the BUILTIN form: hipcc's scoreboard must see the Q loads retired here.  Its own counted vmcnt waits do
    // not know about the asm-issued DMA; left to itself it re-waits "for Q" inside the loop with counts that, with a
    // prefetch in flight, drain the DMA in the middle of the QK MFMAs (simm16 0x0F70: vmcnt 0, expcnt 7, lgkmcnt 15).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int prow = pi23(lq);                       // permuted MFMA row of this lane (K key row / V^T d row inside a 32-block)
    const int k_row_off = prow * 256, k_swz = prow & 15;          // + t2*32 rows: (key & 15) unchanged
    const int v_row_off = prow * 128, v_swz = (prow >> 1) & 7;    // + dt*32 rows: ((d>>1)&7) unchanged

    const bool probe = a.clk_probe && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0;
    unsigned long long pc0 = 0, pw0 = 0, pbar = 0;
    if (probe) { pc0 = clock64(); pw0 = wall_clock64(); }
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int key0 = kt * KB;
        unsigned long long pb0 = 0;
        if (probe) pb0 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt (and, first time, its Q rows)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                      // tile kt complete in LDS; every wave is done with tile kt-1
        __builtin_amdgcn_sched_barrier(0);
        if (probe) pbar += clock64() - pb0;
        if (kt + 1 < kt_hi) issue(kt + 1);
        const char* Ks = smem + (kt & 1) * BUF;
        const char* Vs = Ks + 16384;
        // band: tiles wholly outside this wave's 32 query rows are skipped (wave-uniform)
        const bool in_band = (key0 - (qw0 + 31) <= win) && (qw0 - (key0 + KB - 1) <= win);
        if (!wave_live || !in_band) continue;

        // ---- S^T = K Q^T for both 32-key halves: all 16 K fragments are read in one batch, then two independent MFMA chains
        bf16x8 fr[16];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                fr[t2 * 8 + ks] = as_bf16x8(*reinterpret_cast<const uint4*>(Ks + t2 * 8192 + k_row_off + (((ks * 2 + half) ^ k_swz) << 4)));
        __builtin_amdgcn_sched_barrier(0);  // keep the 16 reads one batch (hipcc otherwise drips them between the MFMAs behind lgkmcnt(0))
        f32x16 s[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            s[0] = mfma32(fr[ks], qf[ks], s[0]);
            s[1] = mfma32(fr[8 + ks], qf[ks], s[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragments (same registers; issued now so that their LDS latency hides under the softmax arithmetic):
        // MFMA step (t2, t) covers keys t2*32 + 16t .. +15; lane-half `half` supplies +8*half .. +7 = registers 8t..8t+7 of s[t2]
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    fr[(t2 * 2 + t) * 4 + dt] =
                        as_bf16x8(*reinterpret_cast<const uint4*>(Vs + dt * 4096 + v_row_off + (((t2 * 4 + 2 * t + half) ^ v_swz) << 4)));
        // lane owns query qrow; register r of s[t2] is key key0 + t2*32 + 16*(r>>3) + 8*half + (r&7)
        const bool interior = (key0 + KB <= skv) && (qw0 + 31 - key0 <= win) && (key0 + KB - 1 - qw0 <= win);
        if (!interior) {
            const int kb = key0 + 8 * half;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb + t2 * 32 + 16 * (r >> 3) + (r & 7);
                    const bool ok = (key < skv) & ((unsigned)(qrow - key + win) <= (unsigned)(2 * win));
                    s[t2][r] = ok ? s[t2][r] : -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * scale_log2);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t2][r], scale_log2, -m_use));
                s[t2][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint4 pb;
                pb.x = pack_bf2(s[t2][8 * t + 0], s[t2][8 * t + 1]);
                pb.y = pack_bf2(s[t2][8 * t + 2], s[t2][8 * t + 3]);
                pb.z = pack_bf2(s[t2][8 * t + 4], s[t2][8 * t + 5]);
                pb.w = pack_bf2(s[t2][8 * t + 6], s[t2][8 * t + 7]);
                const bf16x8 pf = as_bf16x8(pb);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = mfma32(fr[(t2 * 2 + t) * 4 + dt], pf, o[dt]);
            }
    }

    if (probe) {
        g_attn_probe[0] = clock64() - pc0;
        g_attn_probe[1] = wall_clock64() - pw0;
        g_attn_probe[2] = pbar;
        g_attn_probe[3] = (unsigned long long)(kt_hi - kt_lo);
        g_attn_probe[4] = pc0 - t_entry;
    }
    // O[q][d] = O^T[d][q] / l; register r of o[dt] is d = dt*32 + 16*(r>>3) + 8*half + (r&7)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow < a.Sq && l_tot == 0.f) {
        // no valid key at all (only possible with kv_len): the reference's finfo.min mask makes this row uniform over ALL keys
        bf16_t* op = a.out + (long)n * a.o_seq_stride + (long)qrow * a.o_row_stride + h * 128 + 8 * half;
        const bf16_t* vm = a.vmean + ((long)n * a.Hkv + hkv) * 128 + 8 * half;
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(op + c * 16) = *reinterpret_cast<const uint4*>(vm + c * 16);
    } else if (qrow < a.Sq) {
        bf16_t* op = a.out + (long)n * a.o_seq_stride + (long)qrow * a.o_row_stride + h * 128 + 8 * half;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint4 pk;
                pk.x = pack_bf2(o[dt][8 * g + 0] * inv, o[dt][8 * g + 1] * inv);
                pk.y = pack_bf2(o[dt][8 * g + 2] * inv, o[dt][8 * g + 3] * inv);
                pk.z = pack_bf2(o[dt][8 * g + 4] * inv, o[dt][8 * g + 5] * inv);
                pk.w = pack_bf2(o[dt][8 * g + 6] * inv, o[dt][8 * g + 7] * inv);
                *reinterpret_cast<uint4*>(op + dt * 32 + 16 * g) = pk;
            }
    }
    if (probe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_attn_probe[5] = clock64() - t_entry; }
}

}  // namespace

int launch_attention(const AttnArgs& a, hipStream_t s) {
    ACE_CHECK(a.N > 0 && a.Sq > 0 && a.Skv > 0, "attention: empty problem");
    ACE_CHECK(!a.use_tab || a.N <= 64, "attention: at most 64 sequences with pointer tables");
    ACE_CHECK(a.Hq % a.Hkv == 0, "attention: Hq % Hkv");
    ACE_CHECK(a.vt_ld % 64 == 0 && a.vt_ld >= ((a.Skv + 63) / 64) * 64, "attention: V^T row stride must be a padded multiple of 64");
    ACE_CHECK(a.q_row_stride % 8 == 0 && a.k_row_stride % 8 == 0 && a.o_row_stride % 4 == 0, "attention: strides");
    const float scale_log2 = a.scale * 1.4426950408889634f;
    // 2 workgroups per CU are resident (VGPR-bound): pick the block height that wastes fewer resident-slot rounds
    auto cost = [&](int qb) {
        const long blocks = (long)((a.Sq + qb - 1) / qb) * a.Hq * a.N;
        return (double)((blocks + 511) / 512) * qb;
    };
    static int force = -1;
    if (force < 0) { const char* e = getenv("ACE355_ATTN_NW"); force = e ? atoi(e) : 0; }
    // measured (metric config, same box): the 96-query variant is 4% SLOWER than 128 despite the better slot quantisation
    // (256 VGPRs + spill); it is opt-in via ACE355_ATTN_NW=3 / =-1 (cost model) for other shapes.
    static int ver = -1;
    if (ver < 0) { const char* e = getenv("ACE355_ATTN"); ver = (e && e[0] == 'v' && e[1] == '2') ? 2 : 3; }
    ACE_CHECK(!a.kv_len || a.vmean, "attention: kv_len needs vmean");
    ACE_CHECK(!a.kv_len || (a.o_row_stride % 8) == 0, "attention: key-padding masks need 16-byte aligned output rows");
    if ((ver == 3 || a.kv_len) && (a.o_row_stride % 8) == 0) {
        // 2 waves per SIMD (256-VGPR budget): 4-wave blocks (128 queries, two workgroups per CU) by default; 8-wave blocks
        // (256 queries, K/V staged once per 256 rows) when the sequence is long enough to fill the chip with them
        static int nw_env = -1;
        if (nw_env < 0) { const char* e = getenv("ACE355_ATTN_NW3"); nw_env = e ? atoi(e) : 0; }
        const long heads = (long)a.Hq * a.N;
        int nw = (a.Sq >= 1024 && heads * ((a.Sq + 255) / 256) >= 512) ? 8 : 4;
        if (nw_env == 4 || nw_env == 8) nw = nw_env;
        const int qbk = nw * 32;
        dim3 grid((a.Sq + qbk - 1) / qbk, a.Hq, a.N);
        static int clk = -1;
        if (clk < 0) { const char* e = getenv("ACE355_ATTN_CLK"); clk = e ? atoi(e) : 0; }
        AttnArgs ap = a;
        ap.clk_probe = clk;
        if (nw == 8) hipLaunchKernelGGL(attn3_kernel<8>, grid, dim3(512), 0, s, ap, scale_log2);
        else hipLaunchKernelGGL(attn3_kernel<4>, grid, dim3(256), 0, s, ap, scale_log2);
        if (clk) {
            unsigned long long hh[8] = {0};
            ACE_HIP(hipStreamSynchronize(s));
            ACE_HIP(hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_attn_probe), sizeof(hh)));
            if (hh[1]) fprintf(stderr, "[ace355 attn clk] N=%d Sq=%d Skv=%d win=%d: %.3f GHz, loop %.0f cycles (%.2f us), %.0f cycles/tile, wait+barrier %.0f cycles/tile; prologue %.0f cycles, whole wave %.0f cycles\n",
                               a.N, a.Sq, a.Skv, a.window, (double)hh[0] / ((double)hh[1] * 10.0), (double)hh[0], (double)hh[1] * 0.01,
                               (double)hh[0] / (double)hh[3], (double)hh[2] / (double)hh[3], (double)hh[4], (double)hh[5]);
        }
        ACE_LAUNCH_CHECK();
        return 0;
    }
    const bool three = force == 3 || (force == -1 && cost(96) < cost(128));
    if (three) {
        dim3 grid((a.Sq + 95) / 96, a.Hq, a.N);
        hipLaunchKernelGGL(attn_kernel<3>, grid, dim3(192), 0, s, a, scale_log2);
    } else {
        static int occ = -1;
        if (occ < 0) { const char* e = getenv("ACE355_ATTN_OCC"); occ = e ? atoi(e) : 2; }
        dim3 grid((a.Sq + 127) / 128, a.Hq, a.N);
        if (occ == 3) hipLaunchKernelGGL((attn_kernel<4, 3>), grid, dim3(256), 0, s, a, scale_log2);
        else hipLaunchKernelGGL((attn_kernel<4, 2>), grid, dim3(256), 0, s, a, scale_log2);
    }
    ACE_LAUNCH_CHECK();
    return 0;
}

}  // namespace ace355
