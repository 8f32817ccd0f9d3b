// attn.hip - bidirectional flash attention for the DiT (full / sliding-band self-attention, cross-attention).
//
// Replaces ALL_ATTENTION_FUNCTIONS[impl] / eager_attention_forward as called from AceStepAttention.forward
// (base.py:351-367) including the dense additive band mask of create_4d_mask (base.py:56-135): the band predicate
// |i-j| <= window is evaluated in-kernel and K/V tiles wholly outside the band are skipped.
//
// gfx950 design (head_dim 128, GQA): see the comment block above attn3_kernel.  The first version of this file
// (register-staged K/V tiles, 8-byte V^T fragment reads, one LDS buffer and two barriers per tile; "attention v2" in DESIGN.md)
// was replaced by it and removed from the source (git history).
#include "common.h"

#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

namespace ace355 {

namespace {

constexpr int KB = 64;   // keys per tile

// ACE355_ATTN_CLK=1 (diagnostic): one wave of workgroup (0,0,0) records shader-clock totals: whole kernel, K loop, and the
// part of the loop spent in the per-tile wait + barrier; launch_attention prints them.
__device__ unsigned long long g_attn_probe[8];

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
__global__ __launch_bounds__(NW * 64, 2) void attn3_kernel(AttnArgs a, float scale_log2, float defer_thr) {
    constexpr int QB = NW * 32;
    constexpr int BUF = 32768;                 // K tile 16 KB | V^T tile 16 KB
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];

    const unsigned long long t_entry = clock64();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ksp = a.kv_split > 1 ? a.kv_split : 1;
    // (integer division runs on the VALU: readfirstlane tells hipcc the quotient is wave-uniform, the DMA helpers need SGPR operands)
    const int qb = __builtin_amdgcn_readfirstlane((int)blockIdx.x / ksp), sp = (int)blockIdx.x - qb * ksp, h = blockIdx.y, n = blockIdx.z;
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
    if (ksp > 1) {  // split-KV: this workgroup's share of the key tiles (possibly empty: it then leaves a neutral partial)
        const int chunk = __builtin_amdgcn_readfirstlane((max(kt_hi - kt_lo, 0) + ksp - 1) / ksp);
        kt_lo = kt_lo + sp * chunk;
        kt_hi = min(kt_hi, kt_lo + chunk);
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
    if constexpr (NW == 4) {
        // Each wave DMAs its own 32 Q rows (256 B each, full lines: a row-per-lane 16-byte load pattern fetches every line
        // eight times) into its 8 KB slice of the K/V buffer that is idle until tile kt_lo+1 is issued, then reads the
        // fragments back.  Own data: this wave's vmcnt is the only ordering needed; the loop's first barrier keeps every
        // other wave's next-tile DMA out of the slice until the fragments are in registers.
        const unsigned qbuf = lds0 + (unsigned)((kt_lo + 1) & 1) * BUF + (unsigned)wave * 8192u;
        const bf16_t* qb = a.q + (long)n * a.q_seq_stride + h * 128;  // uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row_l = 4 * i + (lane >> 4);
            const int j = (lane & 15) ^ (row_l & 15);
            const unsigned voff = (unsigned)(min(qw0 + row_l, a.Sq - 1) * a.q_row_stride + j * 8) * 2u;
            attn_glds16(voff, qb, (unsigned)__builtin_amdgcn_readfirstlane((int)(qbuf + (unsigned)i * 1024u)));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const char* qs = smem + ((kt_lo + 1) & 1) * BUF + wave * 8192 + lq * 256;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = as_bf16x8(*reinterpret_cast<const uint4*>(qs + (((ks * 2 + half) ^ (lq & 15)) << 4)));
        __builtin_amdgcn_s_waitcnt(0xC07F);  // fragments in registers before the loop's barrier releases the slice
    } else {
        const bf16_t* qp = a.q + (long)n * a.q_seq_stride + (long)qrow_c * a.q_row_stride + h * 128 + half * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = as_bf16x8(ldg16(qp + ks * 16));
        // vmcnt(0) in the BUILTIN form: hipcc's scoreboard must see the Q loads retired here.  Its own counted vmcnt waits do
        // not know about the asm-issued DMA; left to itself it re-waits "for Q" inside the loop with counts that, with a
        // prefetch in flight, drain the DMA in the middle of the QK MFMAs (simm16 0x0F70: vmcnt 0, expcnt 7, lgkmcnt 15).
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
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
        // Deferred rescale: the running max only moves when some row's tile max exceeds it by more than 2^8 (wave-uniform
        // decision); otherwise p = 2^(s - m_run) may reach 256, harmless in fp32 / bf16, and the 64-multiply rescale of O
        // plus the alpha bookkeeping are skipped.  (m_run = -inf on a row's first live tile always takes the move branch.)
        const float m_tile = mx * scale_log2;
        const bool move = __any(m_tile - m_run > defer_thr);
        const float m_new = move ? fmaxf(m_run, m_tile) : m_run;
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = move ? __builtin_amdgcn_exp2f(m_run - m_use) : 1.0f;
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
        if (move && __any(alpha != 1.0f)) {
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
    if (ksp > 1) {
        // split-KV partial: row record = [128 x O (relative to m_run) | m_run | l | 2 pad] floats (16-byte aligned rows); attn_merge_kernel finishes the softmax
        if (qrow < a.Sq) {
            float* pr = a.part + ((((long)sp * a.N + n) * a.Hq + h) * a.Sq + qrow) * 132;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float* d8 = pr + dt * 32 + 16 * g + 8 * half;
                    *reinterpret_cast<float4*>(d8 + 0) = make_float4(o[dt][8 * g + 0], o[dt][8 * g + 1], o[dt][8 * g + 2], o[dt][8 * g + 3]);
                    *reinterpret_cast<float4*>(d8 + 4) = make_float4(o[dt][8 * g + 4], o[dt][8 * g + 5], o[dt][8 * g + 6], o[dt][8 * g + 7]);
                }
            if (half == 0) *reinterpret_cast<float2*>(pr + 128) = make_float2(m_run, l_tot);
        }
        return;
    }
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

// Split-KV merge: out[n][row][h*128 + d] = sum_s 2^(m_s - M) O_s[d] / sum_s 2^(m_s - M) l_s, M = max_s m_s (m in log2 units, as the
// kernels keep it).  32 lanes per row (4 d each), parts summed in part order.
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ part, int nsplit, int N, int Hq, int Sq, bf16_t* __restrict__ out,
                                                         long o_seq_stride, int o_row_stride) {
    const long unit = (long)blockIdx.x * 8 + (threadIdx.x >> 5);   // (n, h, row)
    const int c = threadIdx.x & 31;
    if (unit >= (long)N * Hq * Sq) return;
    const int row = (int)(unit % Sq);
    const int h = (int)((unit / Sq) % Hq);
    const int n = (int)(unit / ((long)Sq * Hq));
    const long stride = (long)N * Hq * Sq * 132;
    const float* pr = part + unit * 132;
    float M = -INFINITY;
    for (int s2 = 0; s2 < nsplit; ++s2) M = fmaxf(M, pr[s2 * stride + 128]);
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    for (int s2 = 0; s2 < nsplit; ++s2) {
        const float m = pr[s2 * stride + 128];
        const float w = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);
        const float4 v = *reinterpret_cast<const float4*>(pr + s2 * stride + 4 * c);
        acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        l += w * pr[s2 * stride + 129];
    }
    const float inv = 1.f / l;
    uint2 pk;
    pk.x = pack_bf2(acc.x * inv, acc.y * inv);
    pk.y = pack_bf2(acc.z * inv, acc.w * inv);
    *reinterpret_cast<uint2*>(out + (long)n * o_seq_stride + (long)row * o_row_stride + h * 128 + 4 * c) = pk;
}

// ------------------------------------------------------------------------------------------------ GQA-shared kernel
// attn3_kernel stages every K / V^T tile once per (q-block, Q HEAD): with Hq = 2 Hkv each tile is DMA'd twice, and at the
// metric shape (Sq = 375 -> 12 wave tiles of 32 rows per (sequence, head), 3072 wave tiles per launch) its two 4-wave
// workgroups per CU give 2048 wave slots: 1.5 rounds, the second one half empty, and every workgroup pays the cold-start
// prologue again.  Here one workgroup owns (sequence, KV head, q-block of 32*NWH rows) and computes BOTH q heads of the group
// off one K / V^T tile: 2*NWH waves (waves [0, NWH) head 2g, waves [NWH, 2*NWH) head 2g+1).  NWH = 6 -> 12 waves, THREE per
// SIMD, 192 query rows: 2 x 8 x 16 = 256 workgroups at the metric = one per CU, one round, half the K / V traffic.
// Three waves per SIMD leave 168 VGPRs, which the 32-register Q fragment set of attn3_kernel does not fit beside O (64),
// S (32) and a useful batch of K fragments: Q lives in LDS instead (each wave DMAs its own 32 rows into a private 8 KB
// slice once) and its fragments are re-read per tile in the same batches as the K fragments; K / V^T fragments are read in
// batches of 8 (half of a tile's contraction) instead of 16.  LDS: 2 x 32 KB K / V^T stages + 2*NWH x 8 KB of Q = 160 KB
// at NWH = 6.  Tile images, swizzles, the row permutation, softmax and epilogue are attn3_kernel's.
template <int NWH>
__global__ __launch_bounds__(NWH * 128, (2 * NWH + 3) / 4) void attn_gqa_kernel(AttnArgs a, float scale_log2, float defer_thr) {
    constexpr int NW = 2 * NWH;
    constexpr int QB = NWH * 32;
    constexpr int BUF = 32768;                  // K tile 16 KB | V^T tile 16 KB
    constexpr int NPI = (32 + NW - 1) / NW;     // DMA pieces per wave per tile (32 pieces of 1 KB)
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF + NW * 8192];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (q-block, KV head, sequence).  Block b runs on XCD b % 8 (observed; speed only): the KV head picks the XCD, so
    // that every workgroup reading one head's K / V^T - the q-blocks of a sequence, and for cross-attention every sequence
    // that attends the same condition slot - shares ONE private L2 (self-attention at the metric: 98 -> 74 MB per launch
    // leave the L2s; cross-attention: the 0.4 MB of a head's keys are fetched once per launch instead of once per workgroup).
    int qb, hkv, n;
    {
        const int nqb = (a.Sq + QB - 1) / QB;
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        if ((a.Hkv & 7) == 0) {
            const int hg = a.Hkv >> 3;
            hkv = xcd + 8 * (j % hg);
            const int rest = j / hg;
            qb = rest % nqb;
            n = rest / nqb;
        } else {  // generic order (grid is padded to a multiple of 8 by the launcher: surplus ids leave)
            hkv = id % a.Hkv;
            const int rest = id / a.Hkv;
            qb = rest % nqb;
            n = rest / nqb;
        }
        if (n >= a.N) return;
    }
    const unsigned long long t_entry = a.clk_probe ? clock64() : 0ull;
    const int hw = wave / NWH, wr = wave - hw * NWH;   // q head inside the group, 32-row tile inside the block
    const int h = hkv * 2 + hw;
    const int q0 = qb * QB;
    const int lq = lane & 31, half = lane >> 5;
    const int qw0 = q0 + wr * 32;
    const int qrow = qw0 + lq;
    const int win = a.window < 0 ? (1 << 28) : a.window;
    const bool wave_live = qw0 < a.Sq;
    const int skv = a.kv_len ? a.kv_len[n] : a.Skv;

    int kt_lo = 0, kt_hi = (skv + KB - 1) / KB;
    if (a.window >= 0) {
        kt_lo = max(0, q0 - a.window) / KB;
        kt_hi = min(kt_hi, (min(skv - 1, q0 + QB - 1 + a.window)) / KB + 1);
    }
    const bf16_t* kbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.k_tab[n]) : a.k + (long)n * a.k_seq_stride) + (long)hkv * a.k_head_stride;
    const bf16_t* vbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.vt_tab[n]) : a.vt + (long)n * a.vt_seq_stride) + (long)hkv * a.vt_head_stride;

    // DMA piece p = wave + NW*i (p < 32): p < 16 = K rows 4p..4p+3, else V^T rows 8(p-16)..+7 (see attn3_kernel)
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    unsigned p_voff[NPI], p_voff_tail[NPI];
    const int tail_key0 = ((a.Skv - 1) / KB) * KB;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = wave + NW * i;
        if (p < 16) {
            const int key = 4 * p + (lane >> 4);
            const int j = (lane & 15) ^ (key & 15);
            p_voff[i] = (unsigned)(key * a.k_row_stride + j * 8) * 2u;
            p_voff_tail[i] = (unsigned)((min(tail_key0 + key, a.Skv - 1) - tail_key0) * a.k_row_stride + j * 8) * 2u;
        } else {
            const int d = 8 * (p - 16) + (lane >> 3);
            const int jv = (lane & 7) ^ ((d >> 1) & 7);
            p_voff[i] = p_voff_tail[i] = (unsigned)(d * a.vt_ld + jv * 8) * 2u;
        }
    }
    auto issue_piece = [&](int kt, int i) {  // this wave's i-th DMA instruction of tile kt
        const int key0 = kt * KB;
        const unsigned bb = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)(kt & 1) * BUF));   // (uniform: the tile range comes from a load)
        const bf16_t* kb_s = kbase + (long)key0 * a.k_row_stride;  // uniform
        const bf16_t* vb_s = vbase + key0;
        const bool tail = key0 + KB > a.Skv;
        const int p = wave + NW * i;  // wave-uniform
        if (p < 16) attn_glds16(tail ? p_voff_tail[i] : p_voff[i], kb_s, bb + (unsigned)p * 1024u);
        else if (p < 32) attn_glds16(p_voff[i], vb_s, bb + (unsigned)p * 1024u);
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NPI; ++i) issue_piece(kt, i);
    };

    // this wave's 32 Q rows -> its private slice (full 256-byte lines, XOR-swizzled chunks), once
    const unsigned qbuf = lds0 + 2u * BUF + (unsigned)wave * 8192u;
    {
        const bf16_t* qsrc = a.q + (long)n * a.q_seq_stride + h * 128;  // uniform
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row_l = 4 * i + (lane >> 4);
            const int j = (lane & 15) ^ (row_l & 15);
            const unsigned voff = (unsigned)(min(qw0 + row_l, a.Sq - 1) * a.q_row_stride + j * 8) * 2u;
            attn_glds16(voff, qsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(qbuf + (unsigned)i * 1024u)));
        }
    }
    if (kt_lo < kt_hi) issue(kt_lo);

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int prow = pi23(lq);
    const int k_row_off = prow * 256, k_swz = prow & 15;
    const int v_row_off = prow * 128, v_swz = (prow >> 1) & 7;
    const char* qs = smem + 2 * BUF + wave * 8192 + lq * 256;
    const int q_swz = lq & 15;

    const bool probe = a.clk_probe && blockIdx.x == 0 && tid == 0;
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
        // The next tile's DMA is NOT issued here: an LDS-DMA instruction takes 60-185 cycles to issue, and with every wave doing
        // it right behind the barrier the matrix pipe sat idle (ablation: 6.4 k -> 4.0 k cycles per tile without barrier + DMA).
        // The pieces ride between the Q K^T batches below (safe any time after the barrier: the target stage was last read in
        // tile kt-1); waves that skip this tile issue theirs at once.
        const bool prefetch = kt + 1 < kt_hi;

        const char* Ks = smem + (kt & 1) * BUF;
        const char* Vs = Ks + 16384;
        const bool in_band = (key0 - (qw0 + 31) <= win) && (qw0 - (key0 + KB - 1) <= win);
        if (!wave_live || !in_band) {
            if (prefetch) issue(kt + 1);
            continue;
        }
        // The 20 fragment addresses of a tile ((slot ^ swizzle) << 4 cannot fold into an immediate) are loop invariant and hipcc
        // hoists them all: 20 VGPRs that at a 168-register budget spill.  Re-materialised per tile (1 VALU each) instead: the
        // swizzle terms go through an opaque asm so that nothing derived from them can leave the loop.
        // (slot ^ swz) << 4 with slot = even | half  ==  (even << 4) ^ ((half ^ swz) << 4): one XOR with a constant per fragment
        int kx = (half ^ k_swz) << 4, qx = (half ^ q_swz) << 4, vx = (half ^ v_swz) << 4;
        asm volatile("" : "+v"(kx), "+v"(qx), "+v"(vx));

        // ---- S^T = K Q^T, contraction in batches of KSB 16-wide steps: 2*KSB K fragments + KSB Q fragments per batch
        // (KSB = 2: 24 fragment registers beside O (64) and S (32) at three waves per SIMD, and four batches to hang the DMA on)
        constexpr int KSB = 2;
        f32x16 s[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t2][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8 / KSB; ++kb) {
            bf16x8 fk[2 * KSB], fq[KSB];
#pragma unroll
            for (int u = 0; u < KSB; ++u) {
                const int ks = kb * KSB + u;
                fq[u] = as_bf16x8(*reinterpret_cast<const uint4*>(qs + ((ks * 32) ^ qx)));
                fk[u] = as_bf16x8(*reinterpret_cast<const uint4*>(Ks + k_row_off + ((ks * 32) ^ kx)));
                fk[KSB + u] = as_bf16x8(*reinterpret_cast<const uint4*>(Ks + 8192 + k_row_off + ((ks * 32) ^ kx)));
            }
            __builtin_amdgcn_sched_barrier(0);  // one batch of reads, then its MFMAs
#pragma unroll
            for (int u = 0; u < KSB; ++u) {
                s[0] = mfma32(fk[u], fq[u], s[0]);
                s[1] = mfma32(fk[KSB + u], fq[u], s[1]);
            }
            if (kb < NPI) {  // one or two DMA instructions of the next tile behind this batch's MFMAs
                __builtin_amdgcn_sched_barrier(0);
                if (prefetch) {
                    issue_piece(kt + 1, kb);
                    if (kb + 4 < NPI) issue_piece(kt + 1, kb + 4);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        static_assert(NPI <= 2 * (8 / KSB), "at most two DMA pieces per Q K^T batch");
        __builtin_amdgcn_sched_barrier(0);
        // first half of the V^T fragments (keys 0..31 of the tile) requested before the softmax arithmetic
        bf16x8 fv[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                fv[t * 4 + dt] = as_bf16x8(*reinterpret_cast<const uint4*>(Vs + dt * 4096 + v_row_off + ((t * 32) ^ vx)));
        const bool interior = (key0 + KB <= skv) && (qw0 + 31 - key0 <= win) && (key0 + KB - 1 - qw0 <= win);
        if (!interior) {
            const int kb0 = key0 + 8 * half;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kb0 + t2 * 32 + 16 * (r >> 3) + (r & 7);
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
        const float m_tile = mx * scale_log2;
        const bool move = __any(m_tile - m_run > defer_thr);
        const float m_new = move ? fmaxf(m_run, m_tile) : m_run;
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = move ? __builtin_amdgcn_exp2f(m_run - m_use) : 1.0f;
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
        if (move && __any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
        // ---- O^T += V^T P^T: keys 0..31 with the fragments already here, the second half's reads issued under those MFMAs
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            bf16x8 pf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint4 pb;
                pb.x = pack_bf2(s[t2][8 * t + 0], s[t2][8 * t + 1]);
                pb.y = pack_bf2(s[t2][8 * t + 2], s[t2][8 * t + 3]);
                pb.z = pack_bf2(s[t2][8 * t + 4], s[t2][8 * t + 5]);
                pb.w = pack_bf2(s[t2][8 * t + 6], s[t2][8 * t + 7]);
                pf[t] = as_bf16x8(pb);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = mfma32(fv[t * 4 + dt], pf[t], o[dt]);
            if (t2 == 0) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        fv[t * 4 + dt] = as_bf16x8(*reinterpret_cast<const uint4*>(Vs + dt * 4096 + v_row_off + ((64 + t * 32) ^ vx)));
            }
        }
    }

    if (probe) {
        g_attn_probe[0] = clock64() - pc0;
        g_attn_probe[1] = wall_clock64() - pw0;
        g_attn_probe[2] = pbar;
        g_attn_probe[3] = (unsigned long long)(kt_hi - kt_lo);
        g_attn_probe[4] = pc0 - t_entry;
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qrow < a.Sq && l_tot == 0.f) {
        bf16_t* op = a.out + (long)n * a.o_seq_stride + (long)qrow * a.o_row_stride + h * 128 + 8 * half;
        const bf16_t* vm = a.vmean + ((long)n * a.Hkv + hkv) * 128 + 8 * half;
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<uint4*>(op + c * 16) = *reinterpret_cast<const uint4*>(vm + c * 16);
    } else if (a.out_q) {
        // MXFP8 output: block dt of this head = columns 32 dt .. +31 = this lane's 16 values (g = 0, 1) + its partner's (lane ^ 32);
        // the values quantised are the bf16-rounded outputs, so this equals attention -> mx_quant_kernel bit for bit.  Every lane
        // runs the shuffles; rows past the end just do not store.
        const long grow = (long)n * a.Sq + qrow;
        uint8_t* oq = a.out_q + grow * (a.Hq * 128) + h * 128 + 8 * half;
        uint32_t word = 0;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float f[16];
            float amax = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const uint32_t pk = pack_bf2(o[dt][e] * inv, o[dt][e + 1] * inv);
                f[e] = bf_lo(pk), f[e + 1] = bf_hi(pk);
                amax = fmaxf(amax, fmaxf(fabsf(f[e]), fabsf(f[e + 1])));
            }
            amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
            int sbe = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;
            sbe = sbe < 0 ? 0 : (sbe > 254 ? 254 : sbe);
            const float sinv = __uint_as_float((uint32_t)(254 - sbe) << 23);
            word |= (uint32_t)sbe << (8 * dt);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint32_t w[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float a0 = fminf(fmaxf(f[8 * g + 4 * e + 0] * sinv, -448.f), 448.f), a1 = fminf(fmaxf(f[8 * g + 4 * e + 1] * sinv, -448.f), 448.f);
                    const float a2 = fminf(fmaxf(f[8 * g + 4 * e + 2] * sinv, -448.f), 448.f), a3 = fminf(fmaxf(f[8 * g + 4 * e + 3] * sinv, -448.f), 448.f);
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);
                    w[e] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, pk, true);
                }
                if (qrow < a.Sq) *reinterpret_cast<uint2*>(oq + dt * 32 + 16 * g) = make_uint2(w[0], w[1]);
            }
        }
        if (qrow < a.Sq && half == 0) a.out_scales[(long)h * a.out_pad + grow] = word;
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

// A ROTATED variant of this kernel was built and measured in round 2 and removed again (numbers in DESIGN.md section 5):
// wave group g = wave / 4 ran its tile loop rotated by g segments around the per-tile barrier (group 0: QK SM PV, group 1:
// PV' QK SM, group 2: SM' PV' QK) so that two waves of a SIMD always fed the matrix pipe while the third did softmax arithmetic.
// It needs tiles k-1 and k live with k+1, k+2 in flight, i.e. four stages, which beside 96 KB of Q only fit as 32-key tiles -
// and at 32 keys the per-tile fixed cost (24 instead of 20 fragment reads per 32 keys, twice the barriers, max / ballot / branch
// logic per tile) ate the gain: 3.7 k cycles per 32-key tile against 5.9 k per 64-key tile here (self-attention 34.9 vs 29.7 us,
// cross-attention 44.6 vs 36.7 us).  Static or rotating s_setprio per wave group moved the wait between waves, not the total.
// Ablations of THIS kernel (ACE355_ATTN_CLK bits): softmax arithmetic off 6.4 k -> 5.6 k cycles per tile, barrier + DMA off 4.0 k,
// both off 3.4 k (MFMA issue floor 3.1 k): what is left on the table is the drain / refill of the three in-phase waves of a
// SIMD at every barrier, which only a deeper K / V^T ring (more LDS than 160 KB allows beside Q) would hide.
// A 32-key-tile form of THIS kernel with two 6-wave workgroups per CU (80 KB each, independent barrier domains) was also measured:
// self-attention 42.2 vs 29.7 us, cross-attention 38.8 vs 36.7 us - shorter tiles lose on every count; longer ones do not fit.

// ------------------------------------------------------------------------------------------------ rotated key-split kernel (round 5)
// What attn_gqa_kernel leaves on the table (its ablations above): the waves of a SIMD leave every per-tile barrier IN PHASE - all of
// them Q K^T, then all softmax arithmetic, then all P V - so the matrix pipe idles through every softmax stretch and the VALU through
// every MFMA stretch.  Here the phases are fixed by construction: wave w runs on SIMD w & 3 (waves are dealt round-robin), its GROUP
// is w >> 2, and the three groups walk the same tile sequence rotated by one segment around the (single) per-tile barrier:
//     group 0:  | QK(i)    SM(i)    PV(i)   |          interval i = between barrier i and barrier i + 1
//     group 1:  | PV(i-1)  QK(i)    SM(i)   |
//     group 2:  | SM(i-1)  PV(i-1)  QK(i)   |
// so each SIMD always has two waves on the matrix pipe and one on the VALU.  Tile i-1's V^T is still read in interval i while tile
// i+1 lands.  Rings as built (PD = 2 tiles of DMA in flight): K ring KRING = PD + 1 = 3 stages, V^T ring VRING = PD + 2 = 4 stages of 16 KB each =
// 112 KB, + 48 KB of Q (2 heads x 96 query rows x 256 B) = the whole 160 KB of LDS.  VRING needs the extra stage: the V^T fragments of tile i-1 read by ld_pv
// cross the barrier of interval i WITHOUT an lgkmcnt wait of their own, so the stage they come from must not be a DMA target before interval i + 1.  One workgroup =
// (sequence, KV head, 96-row q-block) as attn_gqa_kernel<3>, but with TWELVE waves - each (q head, 32-row tile) is shared by two
// waves that split every 64-key tile into its two 32-key halves (S is 16 registers instead of 32) and keep separate (m, l, O)
// states, merged through LDS after the loop (each wave finishes half of the head dim).  Per wave and tile: 8 + 8 MFMAs and 16
// exponentials; per SIMD and tile 1536 cycles of MFMA issue against ~1500 of VALU, running side by side instead of in turn.
// Tile images, swizzles, the row permutation and the deferred rescale are attn3_kernel's.
template <int NRT>   // 32-row tiles per q head; waves = 2 heads x NRT x 2 key halves
__global__ __launch_bounds__(NRT * 256, NRT) void attn_rot_kernel(AttnArgs a, float scale_log2, float defer_thr) {
    constexpr int NW = 4 * NRT;
    constexpr int QB = NRT * 32;
    constexpr int KST = 16384;                 // one K tile / one V^T tile
    constexpr int PD = 2;                      // prefetch distance in tiles: interval i requests tile i + PD
    constexpr int KRING = PD + 1, VRING = PD + 2;
    constexpr int VOFF = KRING * KST;          // V^T ring behind the K ring
    constexpr int QOFF = (KRING + VRING) * KST;   // Q slices behind both rings
    constexpr int NPI = (32 + NW - 1) / NW;    // DMA pieces per wave per tile
    constexpr int SCR_ML = NW * 8192;          // merge scratch: NW x 8 KB of O halves, then NW x 512 B of (m, l)
    static_assert(QOFF + 2 * NRT * 8192 >= SCR_ML + NW * 512, "merge scratch must fit the loop's LDS");
    __shared__ __attribute__((aligned(16))) char smem[QOFF + 2 * NRT * 8192];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qb, hkv, n;
    {
        const int nqb = (a.Sq + QB - 1) / QB;
        const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
        if ((a.Hkv & 7) == 0) {
            const int hg = a.Hkv >> 3;
            hkv = xcd + 8 * (j % hg);
            const int rest = j / hg;
            qb = rest % nqb;
            n = rest / nqb;
        } else {
            hkv = id % a.Hkv;
            const int rest = id / a.Hkv;
            qb = rest % nqb;
            n = rest / nqb;
        }
        if (n >= a.N) return;
    }
    const unsigned long long t_entry = a.clk_probe ? clock64() : 0ull;
    const int grp = wave >> 2;                                   // rotation group (waves w, w + 4, w + 8 share SIMD w & 3)
    const int hw = wave / (2 * NRT), rr = wave - hw * 2 * NRT;   // q head inside the group
    const int wr = rr >> 1, kh = rr & 1;                         // 32-row tile inside the block, key half of every tile
    const int pair = hw * NRT + wr;                              // Q slice shared by the two key halves
    const int h = hkv * 2 + hw;
    const int q0 = qb * QB;
    const int lq = lane & 31, half = lane >> 5;
    const int qw0 = q0 + wr * 32;
    const int qrow = qw0 + lq;
    const int win = a.window < 0 ? (1 << 28) : a.window;
    const bool wave_live = qw0 < a.Sq;
    const int skv = __builtin_amdgcn_readfirstlane(a.kv_len ? a.kv_len[n] : a.Skv);   // (uniform: every per-tile predicate below stays on the scalar unit)

    int kt_lo = 0, kt_hi = (skv + KB - 1) / KB;
    if (a.window >= 0) {
        kt_lo = max(0, q0 - a.window) / KB;
        kt_hi = min(kt_hi, (min(skv - 1, q0 + QB - 1 + a.window)) / KB + 1);
    }
    kt_lo = __builtin_amdgcn_readfirstlane(kt_lo);
    kt_hi = __builtin_amdgcn_readfirstlane(kt_hi);
    const bf16_t* kbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.k_tab[n]) : a.k + (long)n * a.k_seq_stride) + (long)hkv * a.k_head_stride;
    const bf16_t* vbase = (a.use_tab ? reinterpret_cast<const bf16_t*>(a.vt_tab[n]) : a.vt + (long)n * a.vt_seq_stride) + (long)hkv * a.vt_head_stride;

    // DMA piece p = wave + NW*i (p < 32): p < 16 = K rows 4p..4p+3, else V^T rows 8(p-16)..+7 (see attn3_kernel)
    const unsigned lds0 = (unsigned)(uintptr_t)smem;
    unsigned p_voff[NPI], p_voff_tail[NPI];
    const int tail_key0 = ((a.Skv - 1) / KB) * KB;
#pragma unroll
    for (int i = 0; i < NPI; ++i) {
        const int p = wave + NW * i;
        if (p < 16) {
            const int key = 4 * p + (lane >> 4);
            const int j = (lane & 15) ^ (key & 15);
            p_voff[i] = (unsigned)(key * a.k_row_stride + j * 8) * 2u;
            p_voff_tail[i] = (unsigned)((min(tail_key0 + key, a.Skv - 1) - tail_key0) * a.k_row_stride + j * 8) * 2u;
        } else {
            const int d = 8 * (p - 16) + (lane >> 3);
            const int jv = (lane & 7) ^ ((d >> 1) & 7);
            p_voff[i] = p_voff_tail[i] = (unsigned)(d * a.vt_ld + jv * 8) * 2u;
        }
    }
    auto vstage = [&](int kt) {  // V^T ring slot of tile kt (wave-uniform)
        const int r = kt - kt_lo;
        return __builtin_amdgcn_readfirstlane(r % VRING);
    };
    auto kstage = [&](int kt) {
        const int r = kt - kt_lo;
        return __builtin_amdgcn_readfirstlane(r % KRING);
    };
    auto issue_piece = [&](int kt, int i) {
        const int key0 = kt * KB;
        const bf16_t* kb_s = kbase + (long)key0 * a.k_row_stride;  // uniform
        const bf16_t* vb_s = vbase + key0;
        const bool tail = key0 + KB > a.Skv;
        const int p = wave + NW * i;  // wave-uniform
        if (p < 16) attn_glds16(tail ? p_voff_tail[i] : p_voff[i], kb_s, lds0 + (unsigned)kstage(kt) * KST + (unsigned)p * 1024u);
        else if (p < 32) attn_glds16(p_voff[i], vb_s, lds0 + VOFF + (unsigned)vstage(kt) * KST + (unsigned)(p - 16) * 1024u);
    };

    // the pair's 32 Q rows -> its 8 KB slice (full 256-byte lines, XOR-swizzled chunks): each of the two waves brings four of the rows' eight KB
    {
        const bf16_t* qsrc = a.q + (long)n * a.q_seq_stride + h * 128;  // uniform
        const unsigned qbuf = lds0 + QOFF + (unsigned)pair * 8192u;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            const int i = kh * 4 + i2;
            const int row_l = 4 * i + (lane >> 4);
            const int j = (lane & 15) ^ (row_l & 15);
            const unsigned voff = (unsigned)(min(qw0 + row_l, a.Sq - 1) * a.q_row_stride + j * 8) * 2u;
            attn_glds16(voff, qsrc, (unsigned)__builtin_amdgcn_readfirstlane((int)(qbuf + (unsigned)i * 1024u)));
        }
    }
#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (kt_lo + t < kt_hi) {
#pragma unroll
            for (int i = 0; i < NPI; ++i) issue_piece(kt_lo + t, i);
        }

    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 sreg;          // S^T of this wave's 32 keys x 32 query rows (group 2 carries it across the barrier)
    bf16x8 pf[2];         // P^T as the B operand of the two 16-key MFMA steps (group 1 carries it across the barrier)
#pragma unroll
    for (int r = 0; r < 16; ++r) sreg[r] = 0.f;
    pf[0] = pf[1] = as_bf16x8(make_uint4(0, 0, 0, 0));

    const int prow = pi23(lq);
    const int k_row_off = prow * 256 + kh * 8192, k_swz = prow & 15;
    const int v_row_off = prow * 128, v_swz = (prow >> 1) & 7;
    const char* qs = smem + QOFF + pair * 8192 + lq * 256;
    const int q_swz = lq & 15;

    // a (wave, tile) pair with nothing to do: rows past the end, the wave's 32 keys past the valid keys or wholly outside the band
    auto tile_live = [&](int kt) {
        const int k0 = kt * KB + 32 * kh;
        return wave_live && (k0 < skv) && (k0 - (qw0 + 31) <= win) && (qw0 - (k0 + 31) <= win);
    };

    // (the ablation arms this kernel was tuned with - softmax / DMA / barrier / MFMA / fragment reads off, one at a time - live in tools/r06_attn_abl.patch)
    // Fragment buffers: every wave reads the NEXT segment's fragments while it computes on the current ones (three buffers of 4 x 16 bytes
    // per lane; the rings keep a tile's data valid across the barrier, so a buffer may be filled in one interval and consumed in the next).
    bf16x8 f0[4], f1[4], f2[4];
    auto ld_qk = [&](bf16x8 (&f)[4], int kt, int kb) {   // K fragments f[0..1], Q fragments f[2..3] of contraction steps 2 kb, 2 kb + 1
        int kx = (half ^ k_swz) << 4, qx = (half ^ q_swz) << 4;
        asm volatile("" : "+v"(kx), "+v"(qx));
        const char* Ks = smem + kstage(kt) * KST + k_row_off;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ks = kb * 2 + u;
            f[u] = as_bf16x8(*reinterpret_cast<const uint4*>(Ks + ((ks * 32) ^ kx)));
            f[2 + u] = as_bf16x8(*reinterpret_cast<const uint4*>(qs + ((ks * 32) ^ qx)));
        }
    };
    auto mma_qk = [&](const bf16x8 (&f)[4], bool first) {
        if (first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sreg[r] = 0.f;
        }
        sreg = mfma32(f[0], f[2], sreg);
        sreg = mfma32(f[1], f[3], sreg);
    };
    auto ld_pv = [&](bf16x8 (&f)[4], int kt, int t) {    // V^T fragments of 16-key step t of this wave's 32 keys: f[dt]
        int vx = (half ^ v_swz) << 4;
        asm volatile("" : "+v"(vx));
        const char* Vs = smem + VOFF + vstage(kt) * KST + v_row_off;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            f[dt] = as_bf16x8(*reinterpret_cast<const uint4*>(Vs + dt * 4096 + ((kh * 64 + t * 32) ^ vx)));
    };
    auto mma_pv = [&](const bf16x8 (&f)[4], int t) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = mfma32(f[dt], t ? pf[1] : pf[0], o[dt]);
    };
    auto dma_tile = [&](int kt) {   // this wave's pieces of tile kt (no-op past the last tile)
        if (kt < kt_hi) {
#pragma unroll
            for (int i = 0; i < NPI; ++i) issue_piece(kt, i);
        }
    };
    auto seg_sm = [&](int kt) {   // online softmax step on sreg -> pf, (m_run, l_run), O rescaled when the running max moved
        const int k0 = kt * KB + 32 * kh;
        const bool interior = (k0 + 32 <= skv) && (qw0 + 31 - k0 <= win) && (k0 + 31 - qw0 <= win);
        if (!interior) {
            const int kb0 = k0 + 8 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb0 + 16 * (r >> 3) + (r & 7);
                const bool ok = (key < skv) & ((unsigned)(qrow - key + win) <= (unsigned)(2 * win));
                sreg[r] = ok ? sreg[r] : -INFINITY;
            }
        }
        float mx = sreg[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sreg[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_tile = mx * scale_log2;
        const bool move = __any(m_tile - m_run > defer_thr);
        const float m_new = move ? fmaxf(m_run, m_tile) : m_run;
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = move ? __builtin_amdgcn_exp2f(m_run - m_use) : 1.0f;
        m_run = m_new;
        // (pairs: v_pk_fma_f32 / v_pk_add_f32 halve the plain VALU work around the 16 exponentials)
        f32x2_t psum2 = {0.f, 0.f};
        const f32x2_t sc2 = {scale_log2, scale_log2}, mu2 = {-m_use, -m_use};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2_t x = {sreg[r], sreg[r + 1]};
            const f32x2_t y = __builtin_elementwise_fma(x, sc2, mu2);
            const f32x2_t p = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
            sreg[r] = p[0];
            sreg[r + 1] = p[1];
            psum2 += p;
        }
        l_run = l_run * alpha + (psum2[0] + psum2[1]);
        if (move && __any(alpha != 1.0f)) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint4 pb;
            pb.x = pack_bf2(sreg[8 * t + 0], sreg[8 * t + 1]);
            pb.y = pack_bf2(sreg[8 * t + 2], sreg[8 * t + 3]);
            pb.z = pack_bf2(sreg[8 * t + 4], sreg[8 * t + 5]);
            pb.w = pack_bf2(sreg[8 * t + 6], sreg[8 * t + 7]);
            pf[t] = as_bf16x8(pb);
        }
    };
    // Loads retire in order: before the barrier of tile kt this wave's pieces of tiles <= kt must have landed (first time: its Q rows
    // too), the PD - 1 tiles requested after it may stay in flight (waves 0-7 own three pieces of a tile, waves 8-11 two).
    auto tile_sync = [&](int kt) {
        static_assert(PD == 2 && NW == 12, "the counted waits below are written for one tile in flight and 12 waves");
        if (kt + 1 < kt_hi) {
            if (wave < 8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                      // that tile is complete in LDS; every wave is done with the previous interval
        __builtin_amdgcn_sched_barrier(0);
    };

    const bool probe = a.clk_probe && blockIdx.x == 0 && tid == 0;
    unsigned long long pc0 = 0, pw0 = 0;
    if (probe) { pc0 = clock64(); pw0 = wall_clock64(); }

#define ROT_SB() __builtin_amdgcn_sched_barrier(0)
    // one tile's Q K^T from the point where f0 holds contraction batch 0: batches 1, 2 into f1, f2, batch 3 into f0 once it is free
#define ROT_QK_TAIL(kt)                      \
    do {                                     \
        ld_qk(f1, (kt), 1); ld_qk(f2, (kt), 2); \
        mma_qk(f0, true);                    \
        ROT_SB();                            \
        ld_qk(f0, (kt), 3);                  \
        mma_qk(f1, false); mma_qk(f2, false);\
        ROT_SB();                            \
    } while (0)
    if (grp == 0) {
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            tile_sync(kt);
            const bool lv = tile_live(kt);
            if (lv) ld_qk(f0, kt, 0);
            ROT_SB();
            if (lv) ROT_QK_TAIL(kt);
            dma_tile(kt + PD);
            ROT_SB();
            if (lv) { ld_pv(f1, kt, 0); ld_pv(f2, kt, 1); mma_qk(f0, false); }
            ROT_SB();
            if (lv) seg_sm(kt);
            ROT_SB();
            if (lv) { mma_pv(f1, 0); mma_pv(f2, 1); }
            ROT_SB();
        }
    } else if (grp == 1) {
        bool lvp = false;   // the previous tile's P V is pending (its P^T in pf, its V^T fragments in f1 / f2)
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            tile_sync(kt);
            const bool lv = tile_live(kt);
            if (lv) ld_qk(f0, kt, 0);
            if (lvp) { mma_pv(f1, 0); mma_pv(f2, 1); }
            ROT_SB();
            dma_tile(kt + PD);
            ROT_SB();
            if (lv) ROT_QK_TAIL(kt);
            if (lv) { ld_pv(f1, kt, 0); ld_pv(f2, kt, 1); mma_qk(f0, false); }
            ROT_SB();
            if (lv) seg_sm(kt);
            ROT_SB();
            lvp = lv;
        }
        if (lvp) { mma_pv(f1, 0); mma_pv(f2, 1); }
    } else {
        bool lvp = false;   // the previous tile's softmax + P V are pending (S^T in sreg, V^T fragments in f1 / f2)
        for (int kt = kt_lo; kt < kt_hi; ++kt) {
            tile_sync(kt);
            const bool lv = tile_live(kt);
            dma_tile(kt + PD);   // this group opens the interval on the VALU: its pieces go out first
            ROT_SB();
            if (lv) ld_qk(f0, kt, 0);
            if (lvp) seg_sm(kt - 1);
            ROT_SB();
            if (lvp) { mma_pv(f1, 0); mma_pv(f2, 1); }
            ROT_SB();
            if (lv) ROT_QK_TAIL(kt);
            if (lv) { ld_pv(f1, kt, 0); ld_pv(f2, kt, 1); mma_qk(f0, false); }
            ROT_SB();
            lvp = lv;
        }
        if (lvp) {
            seg_sm(kt_hi - 1);
            ROT_SB();
            mma_pv(f1, 0); mma_pv(f2, 1);
        }
    }
#undef ROT_QK_TAIL
#undef ROT_SB

    if (probe) {
        g_attn_probe[0] = clock64() - pc0;
        g_attn_probe[1] = wall_clock64() - pw0;
        g_attn_probe[2] = 0;
        g_attn_probe[3] = (unsigned long long)(kt_hi - kt_lo);
        g_attn_probe[4] = pc0 - t_entry;
    }

    // ---- merge of the two key halves: wave kh finishes head dims [64 kh, 64 kh + 64) and hands the other half of its O to its partner
    float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (an empty key range leaves the Q pieces in flight: nothing may land in the scratch later)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();   // every wave is out of the rings and the Q slices: the scratch may overwrite them
    // (o[] is indexed with compile-time constants only: a run-time index would move the accumulators to scratch memory)
    auto park = [&](const f32x16& x0, const f32x16& x1) {
        char* mine = smem + wave * 8192 + lane * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *reinterpret_cast<float4*>(mine + c * 1024) = make_float4(x0[4 * c + 0], x0[4 * c + 1], x0[4 * c + 2], x0[4 * c + 3]);
            *reinterpret_cast<float4*>(mine + (4 + c) * 1024) = make_float4(x1[4 * c + 0], x1[4 * c + 1], x1[4 * c + 2], x1[4 * c + 3]);
        }
    };
    if (kh == 0) park(o[2], o[3]);
    else park(o[0], o[1]);
    *reinterpret_cast<float2*>(smem + SCR_ML + wave * 512 + lane * 8) = make_float2(m_run, l_tot);
    __syncthreads();   // (with the LDS fence: the partner reads what was just written)
    const int pw = wave ^ 1;   // partner (the other key half of the same head and row tile)
    const float2 ml_p = *reinterpret_cast<const float2*>(smem + SCR_ML + pw * 512 + lane * 8);
    const float m_all = fmaxf(m_run, ml_p.x);
    const float w_me = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_all);
    const float w_pa = (ml_p.x == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(ml_p.x - m_all);
    l_tot = l_tot * w_me + ml_p.y * w_pa;
    const float inv = 1.f / l_tot;
    const float c_me = w_me * inv, c_pa = w_pa * inv;
    const char* theirs = smem + pw * 8192 + lane * 16;
    if (qrow < a.Sq && l_tot == 0.f) {
        bf16_t* op = a.out + (long)n * a.o_seq_stride + (long)qrow * a.o_row_stride + h * 128 + 64 * kh + 8 * half;
        const bf16_t* vm = a.vmean + ((long)n * a.Hkv + hkv) * 128 + 64 * kh + 8 * half;
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(op + c * 16) = *reinterpret_cast<const uint4*>(vm + c * 16);
    } else {
        bf16_t* op = a.out + (long)n * a.o_seq_stride + (long)qrow * a.o_row_stride + h * 128 + 64 * kh + 8 * half;
        auto finish = [&](const f32x16& x, int d2) {   // head dims 64 kh + 32 d2 .. + 31 of this lane's row
            float v[16];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 t4 = *reinterpret_cast<const float4*>(theirs + (d2 * 4 + c) * 1024);
                v[4 * c + 0] = x[4 * c + 0] * c_me + t4.x * c_pa;
                v[4 * c + 1] = x[4 * c + 1] * c_me + t4.y * c_pa;
                v[4 * c + 2] = x[4 * c + 2] * c_me + t4.z * c_pa;
                v[4 * c + 3] = x[4 * c + 3] * c_me + t4.w * c_pa;
            }
            if (qrow < a.Sq) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    uint4 pk;
                    pk.x = pack_bf2(v[8 * g + 0], v[8 * g + 1]);
                    pk.y = pack_bf2(v[8 * g + 2], v[8 * g + 3]);
                    pk.z = pack_bf2(v[8 * g + 4], v[8 * g + 5]);
                    pk.w = pack_bf2(v[8 * g + 6], v[8 * g + 7]);
                    *reinterpret_cast<uint4*>(op + d2 * 32 + 16 * g) = pk;
                }
            }
        };
        if (kh == 0) { finish(o[0], 0); finish(o[1], 1); }
        else { finish(o[2], 0); finish(o[3], 1); }
    }
    if (probe) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); g_attn_probe[5] = clock64() - t_entry; }
}

}  // namespace

static int gqa_nwh(const AttnArgs& a) {  // which attn_gqa_kernel instantiation launch_attention picks (0: attn3_kernel)
    static int gqa_env = -2;
    if (gqa_env == -2) { const char* e = getenv("ACE355_ATTN_GQA"); gqa_env = e ? atoi(e) : -1; }
    if (a.Hq != 2 * a.Hkv || gqa_env == 0) return 0;
    auto units = [&](int nwh) { return (long)a.N * a.Hkv * ((a.Sq + 32 * nwh - 1) / (32 * nwh)); };
    if (gqa_env == 3 || gqa_env == 4 || gqa_env == 6) return gqa_env;
    static int cus_env = -1;   // AttnArgs::cu_slots, else ACE355_MAX_WGS (gemm.hip): plan for this many CUs instead of 256
    if (cus_env < 0) { const char* e = getenv("ACE355_MAX_WGS"); cus_env = e ? atoi(e) : 256; if (cus_env < 8 || cus_env > 256) cus_env = 256; }
    const int cus = (a.cu_slots >= 8 && a.cu_slots <= 256) ? a.cu_slots : cus_env;
    if (units(6) >= 224 * cus / 256) return 6;
    if (units(4) >= 224 * cus / 256) return 4;
    if (units(3) >= 128 * cus / 256) return 3;
    return 0;
}
bool attention_mx_out_ok(const AttnArgs& a) { return gqa_nwh(a) != 0 && !a.kv_len; }

int launch_attention(const AttnArgs& a, hipStream_t s) {
    ACE_CHECK(a.N > 0 && a.Sq > 0 && a.Skv > 0, "attention: empty problem");
    ACE_CHECK(!a.use_tab || a.N <= 64, "attention: at most 64 sequences with pointer tables");
    ACE_CHECK(a.Hq % a.Hkv == 0, "attention: Hq % Hkv");
    ACE_CHECK(a.vt_ld % 64 == 0 && a.vt_ld >= ((a.Skv + 63) / 64) * 64, "attention: V^T row stride must be a padded multiple of 64");
    ACE_CHECK(a.q_row_stride % 8 == 0 && a.k_row_stride % 8 == 0 && a.o_row_stride % 4 == 0, "attention: strides");
    const float scale_log2 = a.scale * 1.4426950408889634f;
    ACE_CHECK(!a.kv_len || a.vmean, "attention: kv_len needs vmean");
    ACE_CHECK((a.o_row_stride % 8) == 0, "attention: output rows must be 16-byte aligned");
    // 2 waves per SIMD (256-VGPR budget): 4-wave blocks (128 queries, two workgroups per CU) by default; 8-wave blocks
    // (256 queries, K/V staged once per 256 rows) when the sequence is long enough to fill the chip with them
    static float thr = -1.f;  // deferred-rescale threshold in log2 units (ACE355_ATTN_DEFER; 0 = rescale whenever a max moves)
    if (thr < 0.f) { const char* e = getenv("ACE355_ATTN_DEFER"); thr = e ? (float)atof(e) : 8.0f; }
    static int nw_env = -1, clk = -1;
    if (nw_env < 0) { const char* e = getenv("ACE355_ATTN_NW"); nw_env = e ? atoi(e) : 0; }
    if (clk < 0) { const char* e = getenv("ACE355_ATTN_CLK"); clk = e ? atoi(e) : 0; }
    // GQA-shared kernel (Hq = 2 Hkv): one workgroup per (sequence, KV head, 32*NWH query rows), both q heads off one K / V^T
    // tile.  NWH = the largest of {6, 4, 3} whose grid still gives every CU a workgroup (ACE355_ATTN_GQA=0: attn3_kernel only,
    // =3/4/6 pins NWH); attn3_kernel keeps the small problems (fewer than 128 such workgroups) and other group sizes.
    ACE_CHECK(!a.out_q || (attention_mx_out_ok(a) && a.out_scales && a.o_seq_stride == (long)a.Sq * a.Hq * 128), "attention: MXFP8 output needs the GQA kernel and a dense [N*Sq, Hq*128] output");
    {
        auto units = [&](int nwh) { return (long)a.N * a.Hkv * ((a.Sq + 32 * nwh - 1) / (32 * nwh)); };
        const int nwh = gqa_nwh(a);
        // rotated key-split kernel (12 waves per 96-row block): where the 6-wave GQA kernel would run (the metric's cross-attention),
        // ACE355_ATTN_ROT=2 also where the 192- / 128-row kernels would (A/B of the self-attention launches), =0 never
        static int rot_env = -1;
        if (rot_env < 0) { const char* e = getenv("ACE355_ATTN_ROT"); rot_env = e ? atoi(e) : 1; }
        // (the key-split kernel merges two partial softmaxes per row: another fp32 summation order than the one-walk kernels.  Which kernel
        //  a sequence gets depends on how many sequences share the launch, so with ace355_gemm_set_k_rotation(0) - "one summation order
        //  whatever the launch shape" - it stays off: a song alone and inside a batch then take the same per-row arithmetic again)
        if (nwh && !a.out_q && gemm_k_rotation_mode() != 0 && (rot_env >= 2 || (rot_env == 1 && nwh == 3))) {
            AttnArgs ap = a;
            ap.clk_probe = clk;
            const long total = units(3);
            const dim3 grid((unsigned)((total + 7) / 8 * 8));
            hipLaunchKernelGGL((attn_rot_kernel<3>), grid, dim3(768), 0, s, ap, scale_log2, thr);
            ACE_LAUNCH_CHECK();
            if (clk) {
                unsigned long long hh[8] = {0};
                ACE_HIP(hipStreamSynchronize(s));
                ACE_HIP(hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_attn_probe), sizeof(hh)));
                if (hh[1]) fprintf(stderr, "[ace355 attn-rot clk] N=%d Sq=%d Skv=%d win=%d: %.3f GHz, loop %.0f cycles (%.2f us), %.0f cycles/tile; prologue %.0f cycles, whole wave %.0f cycles\n",
                                   a.N, a.Sq, a.Skv, a.window, (double)hh[0] / ((double)hh[1] * 10.0), (double)hh[0], (double)hh[1] * 0.01,
                                   (double)hh[0] / (double)hh[3], (double)hh[4], (double)hh[5]);
            }
            return 0;
        }
        if (nwh) {
            AttnArgs ap = a;
            ap.clk_probe = clk;  // bit 0: probe + print; bits 1..3: ablations (attn_gqa_kernel)
            const long total = units(nwh);
            const dim3 grid((unsigned)((total + 7) / 8 * 8));  // 1-D: the kernel decodes (q-block, KV head, sequence) XCD-aware
            if (nwh == 6) hipLaunchKernelGGL(attn_gqa_kernel<6>, grid, dim3(768), 0, s, ap, scale_log2, thr);
            else if (nwh == 4) hipLaunchKernelGGL(attn_gqa_kernel<4>, grid, dim3(512), 0, s, ap, scale_log2, thr);
            else hipLaunchKernelGGL(attn_gqa_kernel<3>, grid, dim3(384), 0, s, ap, scale_log2, thr);
            ACE_LAUNCH_CHECK();
            if (clk) {
                unsigned long long hh[8] = {0};
                ACE_HIP(hipStreamSynchronize(s));
                ACE_HIP(hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_attn_probe), sizeof(hh)));
                if (hh[1]) fprintf(stderr, "[ace355 attn-gqa clk] NWH=%d N=%d Sq=%d Skv=%d win=%d flags=%d: %.3f GHz, loop %.0f cycles (%.2f us), %.0f cycles/tile, wait+barrier %.0f cycles/tile; prologue %.0f cycles\n",
                                   nwh, a.N, a.Sq, a.Skv, a.window, clk, (double)hh[0] / ((double)hh[1] * 10.0), (double)hh[0], (double)hh[1] * 0.01,
                                   (double)hh[0] / (double)hh[3], (double)hh[2] / (double)hh[3], (double)hh[4]);
            }
            return 0;
        }
    }
    const long heads = (long)a.Hq * a.N;
    int nw = (a.Sq >= 1024 && heads * ((a.Sq + 255) / 256) >= 512) ? 8 : 4;
    if (nw_env == 4 || nw_env == 8) nw = nw_env;
    const int qbk = nw * 32;
    const int nqb = (a.Sq + qbk - 1) / qbk;
    AttnArgs ap = a;
    ap.clk_probe = clk;
    // split-KV: few workgroups, each with a long serial walk over the key tiles (batch-1 requests) -> up to 8 workgroups per
    // (q-block, head, sequence), at least two tiles each, about one workgroup per CU in total; needs the caller's scratch and no
    // key-padding mask (the all-masked-row rule of the encoders lives in the one-pass epilogue)
    ap.kv_split = 1;
    {
        static int split_env = -1;
        if (split_env < 0) { const char* e = getenv("ACE355_ATTN_KVSPLIT"); split_env = e ? atoi(e) : 0; }  // 0 heuristic, 1 off, n forced
        const long wgs = (long)nqb * heads;
        int tiles = (a.Skv + KB - 1) / KB;
        if (a.window >= 0) tiles = std::min(tiles, (qbk + 2 * a.window + KB - 1) / KB + 1);
        int sp = 1;
        // (measured at batch 1, S = 375: the 13-tile cross-attention walk 22.2 -> 13.9 + 5.2 us in 4 parts; the 6-tile self-attention
        //  14.1 -> 12.9 + 5.1 us in 2 parts, a loss: the fp32 partials and the merge launch cost ~8 us, so only long walks split)
        // (K rotation mode 0, "one summation order whatever the launch shape": no split either - the merge of partial softmaxes is another order)
        if (nw == 4 && a.part && !a.kv_len && !a.out_q && wgs < 128 && tiles >= 8 && gemm_k_rotation_mode() != 0) {
            while (sp < 8 && wgs * sp * 2 <= 256 && tiles / (sp * 2) >= 2) sp *= 2;
            if (split_env >= 1) sp = split_env;
            if ((long)sp * a.N * a.Hq * a.Sq * 132 > a.part_floats) sp = 1;
        }
        ap.kv_split = sp;
    }
    dim3 grid(nqb * ap.kv_split, a.Hq, a.N);
    if (nw == 8) hipLaunchKernelGGL(attn3_kernel<8>, grid, dim3(512), 0, s, ap, scale_log2, thr);
    else hipLaunchKernelGGL(attn3_kernel<4>, grid, dim3(256), 0, s, ap, scale_log2, thr);
    ACE_LAUNCH_CHECK();
    if (ap.kv_split > 1) {
        const long units = (long)a.N * a.Hq * a.Sq;
        hipLaunchKernelGGL(attn_merge_kernel, dim3((unsigned)((units + 7) / 8)), dim3(256), 0, s, a.part, ap.kv_split, a.N, a.Hq, a.Sq, a.out,
                           a.o_seq_stride, a.o_row_stride);
        ACE_LAUNCH_CHECK();
    }
    if (clk) {
        unsigned long long hh[8] = {0};
        ACE_HIP(hipStreamSynchronize(s));
        ACE_HIP(hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_attn_probe), sizeof(hh)));
        if (hh[1]) fprintf(stderr, "[ace355 attn clk] N=%d Sq=%d Skv=%d win=%d: %.3f GHz, loop %.0f cycles (%.2f us), %.0f cycles/tile, wait+barrier %.0f cycles/tile; prologue %.0f cycles, whole wave %.0f cycles\n",
                           a.N, a.Sq, a.Skv, a.window, (double)hh[0] / ((double)hh[1] * 10.0), (double)hh[0], (double)hh[1] * 0.01,
                           (double)hh[0] / (double)hh[3], (double)hh[2] / (double)hh[3], (double)hh[4], (double)hh[5]);
    }
    return 0;
}

}  // namespace ace355
