// gemm.hip - bf16 MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (fp32 accumulate) with the DiT's fused epilogues.
//
// Replaces the nn.Linear contractions of AceStepDiTLayer (q/k/v/o_proj, base.py:279-282; Qwen3MLP gate/up/down,
// base.py:469), proj_in/proj_out as GEMMs (base.py:1264-1274, 1287-1297) and condition_embedder (base.py:1282).
//
// Two kernels:
//   gemm_kernel     (v1)  register-staged 128x128x64 tile: the simple bring-up / A-B reference (ACE355_GEMM=v1).
//   gemm_sp_kernel  (v4)  the product kernel: A/W tiles HBM -> LDS by DMA (global_load_lds, issued from asm so hipcc does
//                         not drain it), two LDS stages, K loop rotated by half a step, DMA pieces / fragment reads
//                         interleaved one per MFMA, 192x256 / 192x128 (8 waves) or 128/192x128 (4 waves) block tiles,
//                         grouped XCD-local rasterisation, persistent workgroups for multi-round launches.  DESIGN.md
//                         section 5 has the measured ladder; the ablation variants (no-DMA / no-fragment-read / no-barrier),
//                         the non-interleaved schedule and the ping-pong wave-group kernel that produced its numbers were
//                         removed from the source once measured (git history: "ping-pong wave-group kernel", "GEMM v4").
// Round 2 on top of that: an MX-scaled fp8 variant of the same kernel (FP8 template parameter, v_mfma_scale_f32_32x32x64_f8f6f4), the
// folded-RMSNorm hooks of the epilogues (GemmEpilogue nf_* / nc_*: DESIGN.md section 5), split-K of the small-M residual launches in
// part order (turn counters, bit-reproducible) behind a one-time check of the workgroup -> XCD placement it relies on.
// Common: MFMA 16x16x32 bf16 (AccTile below; the bring-up kernel and the MX kernels keep 32x32 forms), K-contiguous operands, LDS image XOR-swizzled at 16-B granularity (slot ^= (row>>1)&7:
// every ds_read_b128 fragment read is bank-conflict free; with DMA the swizzle is applied on the source address).
#include "common.h"

// Compile-time A/B arms of rounds 4-5 (K-loop ablations ACE355_ABL_HOTK / ABL_NODMA, EPI_VEC = 0, EPI_NT, MFMA_ORDER 1 / 2 / 3, MFMA_PAIR 0 / 2, PAIR_PCS,
// WAVE_PAIR) were removed in round 6 once measured (git history up to dd0c588 has them; DESIGN.md sections 12-13 have their numbers).  What is left is
// what ships.  MFMA issue rule of this file (round 6): every v_mfma_f32_16x16x32_bf16 is issued from asm with D == C, and no two
// CONSECUTIVE MFMAs of a wave read the same srcA registers - both were found the hard way.

#include <stdio.h>
#include <type_traits>
#include <stdlib.h>
#include <algorithm>

namespace ace355 {

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

// ACE355_GEMM_CLK=1 (diagnostic): workgroup 0 records shader-clock and 100 MHz wall-clock deltas around its K loop;
// launch_gemm then prints the effective shader clock (DVFS) and the cycles per K-step to stderr.
__device__ unsigned long long g_clk_probe[12];

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }


// Epilogue shared by all mainloops.  Every mainloop issues its MFMAs with the operands SWAPPED (D = Wfrag x Afrag^T), so in
// a 32x32 accumulator tile lane l holds ONE output row m = .. + (l&31) and, per register quad g = r>>2, FOUR consecutive
// columns n = .. + 8g + 4(l>>5) + (r&3).  The stores of an MFMA epilogue are issue-bound (cost per instruction, not per
// byte: a 2-byte-per-lane store pattern cost 7.5 us per 192x256 tile round, 16 % of the GEMM time), so the tile goes
// through a wave-private LDS staging image and leaves as full 128-byte lines, 16 bytes per lane:
//   bf16 modes: pack 4 columns -> ds_write_b64;  fp32 modes: ds_write_b128, one 32-column half (j) at a time;
//   read back row-major (8 lanes x 16 B = one 128-byte row, XOR-swizzled 16-B slots), then global 16-B loads / stores.
// The fp32 residual update (mode 2) issues all its loads for a half before the FMAs and stores.
// Tiles that are partial in N (or a misaligned C) take the scalar path below.
// Accumulator tile of one 32x32 output block per wave, in one of two register layouts (round 4):
//   L16 = false: one v_mfma_f32_32x32x16_bf16 accumulator (16 registers): lane l holds row l & 31, quad q = columns 8q + 4 (l >> 5) .. + 3
//   L16 = true : four v_mfma_f32_16x16x32_bf16 accumulators (4 registers each), quad q = (row half q >> 1, column half q & 1): lane l
//                holds row 16 (q >> 1) + (l & 15), columns 16 (q & 1) + 4 (l >> 4) .. + 3
// (operands swapped in both: D = Wfrag x Afrag^T, so a lane owns ONE output row per quad and four consecutive columns).  The bf16
// kernels use L16: under the power cap the 16x16x32 form sustains 2135-2276 TFLOP/s against 1792-2049 for 32x32x16 on a pure MFMA
// loop (tools/probe/mfma_shape_probe.hip: half the accumulator register traffic per FLOP), same pipe time per FLOP.
typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <bool L16> struct AccTile { f32x16 v; };
template <> struct AccTile<true> { f32x4v q[4]; };
template <bool L16> __device__ __forceinline__ float aq(const AccTile<L16>& t, int q, int e) {
    if constexpr (L16) return t.q[q][e]; else return t.v[4 * q + e];
}
template <bool L16> __device__ __forceinline__ void aq_add(AccTile<L16>& t, int q, int e, float x) {
    if constexpr (L16) t.q[q][e] += x; else t.v[4 * q + e] += x;
}
template <bool L16> __device__ __forceinline__ void acc_zero(AccTile<L16>& t) {
    if constexpr (L16) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) t.q[q][e] = 0.f;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) t.v[r] = 0.f;
    }
}
template <bool L16> __device__ __forceinline__ int q_row(int q, int lane) { return L16 ? (q >> 1) * 16 + (lane & 15) : (lane & 31); }   // row of quad q inside the block
template <bool L16> __device__ __forceinline__ int q_col(int q, int lane) { return L16 ? (q & 1) * 16 + 4 * (lane >> 4) : 8 * q + 4 * (lane >> 5); }   // its first column
template <bool L16> __device__ __forceinline__ int q_rr(int q) { return L16 ? (q >> 1) : 0; }   // which of the lane's rows of the block (L16: two)

// Second half of a bf16 K step: which DMA piece (or -1) rides behind MFMA slot m, and how many piece-free slots precede slot m (those
// carry the next step's fragment reads).  The ND pieces sit on every (NM / ND)-th slot instead of back to back behind the barrier: a
// piece blocks the issuing wave for 60-180 cycles while the address unit of the CU works through all eight waves' pieces (no-DMA
// ablation: 1985 -> 1678 cycles per K step), and spread out the two waves of a SIMD are less often blocked together: 1984 -> 1821
// cycles per K step on the gate|up launch, 1942 -> 1771 on QKV (profiles/r04/r04_dma_placement_sweep.txt: "pat1"); wave-group-specific
// placements (the second wave of each SIMD one slot later / after its reads) were no better and spilled in the persistent kernels.
template <int NM, int ND>
__device__ __forceinline__ constexpr int dma_piece_of_slot(int m) {
    const int S = NM / ND < 1 ? 1 : NM / ND;
    return (m % S == 0 && m / S < ND) ? m / S : -1;
}
template <int NM, int ND>
__device__ __forceinline__ constexpr int dma_free_rank(int m) {
    int r = 0;
    for (int x = 0; x < m; ++x) r += dma_piece_of_slot<NM, ND>(x) < 0 ? 1 : 0;
    return r;
}

template <int RB>
__device__ __forceinline__ int stage_off(int row, int slot) {
    return RB == 128 ? row * 128 + ((slot ^ (row & 7)) << 4) : row * 64 + ((slot ^ ((row >> 1) & 3)) << 4);
}

__device__ __forceinline__ float4 ldf4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldf4_h(const float* p) {   // an old-H row segment of the residual epilogue (read once)
    return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void stf4_h(float* p, const float4& o) {   // the new-H row segment (next read: the next residual GEMM, ~200 MB of traffic later)
    *reinterpret_cast<float4*>(p) = o;
}
__device__ __forceinline__ void stu2_xg(bf16_t* p, const uint2& v) {   // bf16(h * g): read by the next GEMM's DMA through every XCD's L2
    *reinterpret_cast<uint2*>(p) = v;
}
// v + (v of the lane a DPP control selects): cross-lane adds on the VALU, no LDS round trip
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// c += a0 b0; c += a1 b1 as two v_mfma_f32_16x16x32_bf16 BACK TO BACK with D == C on both (gemm_sp_kernel: kstep_pair).  asm, because hipcc gives the first
// of two chained builtins a scratch destination whenever that suits its register allocation, and only the in-place form is chained by the matrix pipe.
__device__ __forceinline__ void mfma16_pair(f32x4v& c, const bf16x8& a0, const bf16x8& b0, const bf16x8& a1, const bf16x8& b1) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x32_bf16 %0, %3, %4, %0" : "+v"(c) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}

// c += a b as ONE v_mfma_f32_16x16x32_bf16 with D == C, from asm.  Round 6: left to the builtin, hipcc gave two MFMAs of the 4-wave
// 128x128 two-stage SwiGLU loop a destination that OVERLAPS A SOURCE operand and differs from the accumulator it reads
// (`v_mfma_f32_16x16x32_bf16 v[26:29], v[2:5], v[26:29], v[34:37]`: D == B != C; LLVM allows it for 4-register destinations) - and that launch
// (M = 288 / 400, N = 12288, K = 2048: the condition encoder's gate|up projection, the one launch of the suite on that instantiation with a long K
// loop) returned different bits on every call, up to 2.6e-2 from fp32 (tools/r06_gemm_determinism.py; profiles/r06_gemm_determinism.txt).  In-place
// accumulation is the only form the kernels need, so it is the only form they issue.
__device__ __forceinline__ void mfma16_inplace(f32x4v& c, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <int MODE, int MT, int NTW, bool ROWS_FULL, bool FOLD = true, bool L16 = false>   // FOLD: the folded-RMSNorm hooks are compiled in (bf16 kernels)
__device__ __forceinline__ void gemm_epilogue_wide(AccTile<L16> (&acc)[MT][NTW], char* __restrict__ stg, void* __restrict__ Cv, int ldc,
                                                   int M, const GemmEpilogue& ep, int mw0, int nw0, int lane, float* xw = nullptr,
                                                   int wave = 0, int wnw = 1, const char* vec = nullptr, int mt0 = 0, int nt0 = 0) {
    // vec (8-wave bf16 kernels, folded-norm consumers): the workgroup's LDS vector area, filled by DMA pieces issued ahead of the tile's
    // first operand tile (gemm_sp_kernel: vec_issue): [0, 2048) the u64 row sums of squares of the tile's rows, [2048, 3072) nc_bias of its
    // columns, [3072, 3584) the head's norm weights (mode 4).  mt0 / nt0 = this wave's first row / column inside the tile.  The row sums
    // were written by memory-side atomics of the producer GEMM: as global loads they opened the epilogue with a fabric round trip, and the
    // per-column vectors cost an L2 round trip per phase; from LDS they are ~64-cycle reads (in registers they spilled: profiles/r05/r05_presq_*).
    constexpr int RPB = L16 ? 2 : 1;   // rows of a 32x32 block one lane holds (AccTile)
    const int lrow = L16 ? (lane & 15) : (lane & 31);   // the lane's row inside its 16- / 32-row group
    const bool eprobe = ep.clk_probe && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0;   // ACE355_GEMM_CLK: phases of this epilogue
    const unsigned long long ep0 = eprobe ? clock64() : 0ull;
    if constexpr (MODE == 0 || MODE == 3 || MODE == 4) {
        // Folded RMSNorm, consumer side: the A operand was h * g (bf16); the row's rstd and the shift's projection complete
        // rmsnorm(h) * g + shift on the fp32 accumulators: t = acc * rs_in[row] + nc_bias[col], applied WHERE a value is consumed (the
        // head-norm's sum of squares, the pack loop).  lane = row (frow), register r = column 8 (r >> 2) + 4 fhalf + (r & 3).  The
        // arithmetic is unconditional (rstd 1, bias 0 with the hooks off: x * 1 + 0 is x) and the accumulators are never written: an
        // in-place update made hipcc move all 96 of them to VGPRs and spill (persistent QKV kernel: 192 B of scratch, epilogue 9.5 k ->
        // 32 k cycles per tile, +19 us per launch even with the hooks OFF).
        float rs_in[MT][RPB];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rr = 0; rr < RPB; ++rr) rs_in[i][rr] = 1.f;
        if (FOLD && ep.nc_rowsq) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rr = 0; rr < RPB; ++rr) {
                    const int m = mw0 + i * 32 + rr * 16 + lrow;
                    const unsigned long long sq = vec ? *reinterpret_cast<const unsigned long long*>(vec + (mt0 + i * 32 + rr * 16 + lrow) * 8)
                                                      : ep.nc_rowsq[ROWS_FULL ? m : min(m, M - 1)];
                    rs_in[i][rr] = rsqrtf((float)(long long)sq * (1.f / 16777216.f) * ep.nc_inv_d + ep.nc_eps);
                }
        }
        auto fold_bias = [&](int j, int g) -> float4 {   // nc_bias of this lane's 4 columns of quad g of block column j; zeros with the hooks off
            float4 bq = {0.f, 0.f, 0.f, 0.f};
            if (FOLD && ep.nc_bias) {
                if (vec) bq = *reinterpret_cast<const float4*>(vec + 2048 + (nt0 + j * 32 + q_col<L16>(g, lane)) * 4);
                else bq = ldf4(ep.nc_bias + nw0 + j * 32 + q_col<L16>(g, lane));
            }
            return bq;
        };
        // mode 4, q / k tiles (workgroup-uniform): per-row sum of squares over the head's 128 columns = this wave's 64 (two
        // half-rows in lanes l and l^32) + the neighbouring wave's 64, swapped through `xw`; then w * (v * rstd) and the
        // rotation of the adjacent (d, d+64) pairs, all on the fp32 accumulators (one bf16 rounding instead of two)
        if constexpr (MODE == 4) {
            if (ep.vt_out && nw0 >= ep.hn_qk_cols) {   // v tile -> V^T (workgroup-uniform: q | k | v boundaries fall on tile edges)
                const int rps = ep.rows_per_seq;
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int rr = 0; rr < RPB; ++rr) {
                        const int m = mw0 + i * 32 + rr * 16 + lrow;
                        const int sq = m / rps, sp = m - sq * rps;
                        bf16_t* base = ep.vt_out + (long)sq * ep.vt_heads * 128 * ep.vt_ld + sp;
                        const bool ok = ROWS_FULL || m < M;
#pragma unroll
                        for (int j = 0; j < NTW; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                if (q_rr<L16>(g) != rr) continue;   // (the quads of this row)
                                const float4 bq = fold_bias(j, g);
                                const int c = nw0 - ep.hn_qk_cols + j * 32 + q_col<L16>(g, lane);   // head * 128 + d of this lane's 4 columns
                                bf16_t* col = base + (long)c * ep.vt_ld;
                                if (ok) {
                                    col[0] = f2bf(__builtin_fmaf(aq<L16>(acc[i][j], g, 0), rs_in[i][rr], bq.x));
                                    col[ep.vt_ld] = f2bf(__builtin_fmaf(aq<L16>(acc[i][j], g, 1), rs_in[i][rr], bq.y));
                                    col[2 * (long)ep.vt_ld] = f2bf(__builtin_fmaf(aq<L16>(acc[i][j], g, 2), rs_in[i][rr], bq.z));
                                    col[3 * (long)ep.vt_ld] = f2bf(__builtin_fmaf(aq<L16>(acc[i][j], g, 3), rs_in[i][rr], bq.w));
                                }
                            }
                    }
                return;
            }
        }
        float rstd[MT][RPB];
        const bool hn = (MODE == 4) && (nw0 < ep.hn_qk_cols);
        const float* hw = nullptr;
        if constexpr (MODE == 4) {
            constexpr int HW = 4 / NTW;  // waves per 128-column head: a pair (64 columns each) or all four N-waves of a 128-wide tile
            if (hn) {
                hw = vec ? reinterpret_cast<const float*>(vec + 3072) : ((nw0 < ep.hn_q_cols) ? ep.hn_wq : ep.hn_wk);
                // One partial per 32-COLUMN BLOCK of the head (4 of them), summed in block order: the sum of a row's 128 squares then has ONE fp32 order whatever
                // the tile form (a pair of 64-column waves or four 32-column waves used to group it differently, and with it the last bit of rstd - the one
                // place where the cross-attention q projection of a single song (64-row tiles) and of a batch (192 x 128 tiles) disagreed; round 6)
#pragma unroll
                for (int j = 0; j < NTW; ++j) {
                    float ss[MT][RPB];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) ss[i][rr] = 0.f;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 bq = fold_bias(j, g);
#pragma unroll
                        for (int i = 0; i < MT; ++i) {
                            const float rs = rs_in[i][q_rr<L16>(g)];
                            const float t0 = __builtin_fmaf(aq<L16>(acc[i][j], g, 0), rs, bq.x), t1 = __builtin_fmaf(aq<L16>(acc[i][j], g, 1), rs, bq.y);
                            const float t2 = __builtin_fmaf(aq<L16>(acc[i][j], g, 2), rs, bq.z), t3 = __builtin_fmaf(aq<L16>(acc[i][j], g, 3), rs, bq.w);
                            ss[i][q_rr<L16>(g)] += t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
                        }
                    }
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int rr = 0; rr < RPB; ++rr) {
                            // the lanes that hold the other columns of this row: l ^ 32 (32x32 layout), l ^ 16 and l ^ 32 (16x16 layout)
                            if constexpr (L16) ss[i][rr] += __shfl_xor(ss[i][rr], 16, 64);
                            ss[i][rr] += __shfl_xor(ss[i][rr], 32, 64);
                            if (lane < (L16 ? 16 : 32)) xw[(wave * NTW + j) * (MT * 32) + i * 32 + rr * 16 + lrow] = ss[i][rr];
                        }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int rr = 0; rr < RPB; ++rr) {
                        float tot = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) tot += xw[((wave & ~(HW - 1)) * NTW + k) * (MT * 32) + i * 32 + rr * 16 + lrow];
                        rstd[i][rr] = rsqrtf(tot * (1.f / 128.f) + ep.hn_eps);
                    }
                if (eprobe) g_clk_probe[9] = clock64() - ep0;    // row scales + head sums of squares exchanged
            }
        }
        // lane's columns c = (nw0 & 127) + j*32 + 8g + 4*fhalf .. c+3 in pair order = dims (t, t+64), (t+1, t+65), t = c / 2.
        // The norm weights are per column (every row-lane of a half reads the same address) and are applied here; the
        // rotation needs cos / sin of (row, pair), which in THIS layout (lane = row) is a 256-byte-strided gather - 32 cache
        // lines per load, +20 us per launch, the whole gain of the fusion - so it is applied in the read-back below, where 8
        // lanes hold one row's 64 columns and read 128 contiguous bytes of the table row.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
        f32x2 hwl[NTW * 4], hwh[NTW * 4];
        const bool rope = (MODE == 4) && ep.hn_cos != nullptr;  // no table: plain column order, norm only (cross-attention q)
        if constexpr (MODE == 4) {
            if (hn) {
                // (the layout test stays outside the unrolled loops: a branch per element fences every load behind the
                //  previous one's wait)
                if (rope) {
#pragma unroll
                    for (int q8 = 0; q8 < NTW * 4; ++q8) {
                        const int c = (nw0 & 127) + (q8 >> 2) * 32 + q_col<L16>(q8 & 3, lane);
                        hwl[q8] = *reinterpret_cast<const f32x2*>(hw + (c >> 1));
                        hwh[q8] = *reinterpret_cast<const f32x2*>(hw + 64 + (c >> 1));
                    }
                } else {
#pragma unroll
                    for (int q8 = 0; q8 < NTW * 4; ++q8) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(hw + (nw0 & 127) + (q8 >> 2) * 32 + q_col<L16>(q8 & 3, lane));
                        hwl[q8] = f32x2{w4[0], w4[2]};
                        hwh[q8] = f32x2{w4[1], w4[3]};
                    }
                }
            }
        }
        static_assert(MODE != 3 || NTW == 2, "SwiGLU pairs two column tiles per wave");
        constexpr int NJ = (MODE == 3) ? 1 : NTW;   // 32-column groups in the staged image
        constexpr int RB = NJ * 64;                 // staged row bytes
        constexpr int J1 = NTW - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = fold_bias(j, g);
                float4 bu = {0.f, 0.f, 0.f, 0.f};
                if constexpr (MODE == 3) bu = fold_bias(J1, g);
                float4 bb = {0.f, 0.f, 0.f, 0.f};
                const int qc = q_col<L16>(g, lane);   // first of this quad's 4 columns inside the 32-column block
                if constexpr (MODE != 4 && MODE != 3) {  // (the head epilogue takes no bias: launch_gemm checks)
                    if (ep.bias) bb = ldf4(ep.bias + nw0 + j * 32 + qc);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int rr = q_rr<L16>(g);
                    const float rs = rs_in[i][rr];
                    float v0, v1, v2, v3;
                    if constexpr (MODE == 3) {  // W rows are interleaved [32 gate | 32 up] per 64: acc[i][0] = gate, acc[i][1] = up
                        v0 = silu_f(__builtin_fmaf(aq<L16>(acc[i][0], g, 0), rs, bq.x)) * __builtin_fmaf(aq<L16>(acc[i][J1], g, 0), rs, bu.x);
                        v1 = silu_f(__builtin_fmaf(aq<L16>(acc[i][0], g, 1), rs, bq.y)) * __builtin_fmaf(aq<L16>(acc[i][J1], g, 1), rs, bu.y);
                        v2 = silu_f(__builtin_fmaf(aq<L16>(acc[i][0], g, 2), rs, bq.z)) * __builtin_fmaf(aq<L16>(acc[i][J1], g, 2), rs, bu.z);
                        v3 = silu_f(__builtin_fmaf(aq<L16>(acc[i][0], g, 3), rs, bq.w)) * __builtin_fmaf(aq<L16>(acc[i][J1], g, 3), rs, bu.w);
                    } else {
                        v0 = __builtin_fmaf(aq<L16>(acc[i][j], g, 0), rs, bq.x) + bb.x; v1 = __builtin_fmaf(aq<L16>(acc[i][j], g, 1), rs, bq.y) + bb.y;
                        v2 = __builtin_fmaf(aq<L16>(acc[i][j], g, 2), rs, bq.z) + bb.z; v3 = __builtin_fmaf(aq<L16>(acc[i][j], g, 3), rs, bq.w) + bb.w;
                        if constexpr (MODE == 4) {
                            if (hn) {
                                const int q8 = j * 4 + g;  // head-norm here (per-column weights broadcast over the rows) ...
                                v0 = hwl[q8].x * (v0 * rstd[i][rr]), v1 = hwh[q8].x * (v1 * rstd[i][rr]);
                                v2 = hwl[q8].y * (v2 * rstd[i][rr]), v3 = hwh[q8].y * (v3 * rstd[i][rr]);
                            }
                        }
                    }
                    uint2 pk;
                    pk.x = pack_bf2(v0, v1);
                    pk.y = pack_bf2(v2, v3);
                    // staged image: 8-column (16-byte) slots; this quad = 4 columns = one half of slot (j * 32 + qc) / 8
                    *reinterpret_cast<uint2*>(stg + stage_off<RB>(i * 32 + q_row<L16>(g, lane), j * 4 + (qc >> 3)) + (qc & 4) * 2) = pk;
                }
            }
        if (eprobe) g_clk_probe[10] = clock64() - ep0;   // ... + staged
        constexpr int LPR = RB / 16, RPI = 64 / LPR;  // lanes per staged row, rows per store instruction
        const int rsub = lane / LPR, slot = lane % LPR;
        bf16_t* out = reinterpret_cast<bf16_t*>(Cv) + (MODE == 3 ? (nw0 >> 1) : nw0) + slot * 8;
        constexpr int NTI = MT * 32 / RPI;  // read-back iterations
        if (MODE == 4 && hn && rope) {
            // ... rotation here: this lane's 8 columns are the pairs t..t+3 of row m (same two roundings as the unfused path).
            // Table rows come in batches of NB iterations, and batch b+1 is requested BEFORE batch b's stores (vmcnt retires in order
            // and counts stores: table loads issued behind a batch of stores wait for those stores' acknowledgements first).  All 12
            // iterations' rows at once = 96 registers: the persistent kernel spills.  Measured neutral at M = 6000 (ABAB 515.0 vs 514.9
            // ms per pass): with two waves per SIMD this epilogue is VALU bound - in-pass phase probe (ACE355_GEMM_CLK), cycles from
            // its start: sums of squares exchanged 4.4 k, normalised + staged 10 k, rotated + stored 16 k (plain store epilogue: 4.8 k).
            constexpr int NB = NTI % 4 == 0 ? 4 : (NTI % 3 == 0 ? 3 : 1);
            constexpr int NBAT = NTI / NB;
            const int pc = ((nw0 & 127) >> 1) + slot * 4;
            f32x4 cs[2][NB], sn[2][NB];
            auto table_rows = [&](int b, f32x4 (&c)[NB], f32x4 (&sv)[NB]) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int row = (b * NB + u) * RPI + rsub;
                    const long po = (long)((mw0 + row) % ep.rows_per_seq) * 64 + pc;
                    c[u] = *reinterpret_cast<const f32x4*>(ep.hn_cos + po);
                    sv[u] = *reinterpret_cast<const f32x4*>(ep.hn_sin + po);
                }
            };
            table_rows(0, cs[0], sn[0]);
#pragma unroll
            for (int b = 0; b < NBAT; ++b) {
                if (b + 1 < NBAT) table_rows(b + 1, cs[(b + 1) & 1], sn[(b + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);   // (keeps the next batch's requests ahead of this batch's stores)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int row = (b * NB + u) * RPI + rsub;
                    const int m = mw0 + row;
                    u32x4e v = *reinterpret_cast<const u32x4e*>(stg + stage_off<RB>(row, slot));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = bf_lo(v[e]), bb2 = bf_hi(v[e]);
                        v[e] = pack_bf2(a * cs[b & 1][u][e] - bb2 * sn[b & 1][u][e], bb2 * cs[b & 1][u][e] + a * sn[b & 1][u][e]);
                    }
                    if (ROWS_FULL || m < M) *reinterpret_cast<u32x4e*>(out + (long)m * ldc) = v;
                }
            }
        } else if (MODE == 3 && ep.mxo_scales) {
            // SwiGLU result straight to MXFP8: the 4 lanes of a staged row hold the wave's 32 output columns = one block
            uint8_t* outq = reinterpret_cast<uint8_t*>(Cv) + (nw0 >> 1) + slot * 8;
            const int kblk = (nw0 >> 1) >> 5;
            uint8_t* scb = reinterpret_cast<uint8_t*>(ep.mxo_scales) + (long)(kblk >> 2) * ep.mxo_pad * 4 + (kblk & 3);
#pragma unroll
            for (int t = 0; t < NTI; ++t) {
                const int row = t * RPI + rsub;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + stage_off<RB>(row, slot));
                const int m = mw0 + row;
                const float f[8] = {bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y), bf_lo(v.z), bf_hi(v.z), bf_lo(v.w), bf_hi(v.w)};
                float amax = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(f[e]));
                amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
                amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
                int sbe = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;
                sbe = sbe < 0 ? 0 : (sbe > 254 ? 254 : sbe);
                const float inv = __uint_as_float((uint32_t)(254 - sbe) << 23);
                uint32_t w[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float a0 = fminf(fmaxf(f[4 * e + 0] * inv, -448.f), 448.f), a1 = fminf(fmaxf(f[4 * e + 1] * inv, -448.f), 448.f);
                    const float a2 = fminf(fmaxf(f[4 * e + 2] * inv, -448.f), 448.f), a3 = fminf(fmaxf(f[4 * e + 3] * inv, -448.f), 448.f);
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, 0, false);
                    w[e] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, pk, true);
                }
                if (ROWS_FULL || m < M) {
                    *reinterpret_cast<uint2*>(outq + (long)m * ldc) = make_uint2(w[0], w[1]);
                    if (slot == 0) scb[(long)m * 4] = (uint8_t)sbe;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NTI; ++t) {
                const int row = t * RPI + rsub;
                const uint4 v = *reinterpret_cast<const uint4*>(stg + stage_off<RB>(row, slot));
                const int m = mw0 + row;
                if (ROWS_FULL || m < M) *reinterpret_cast<uint4*>(out + (long)m * ldc) = v;
            }
        }
    } else {
        constexpr int NT = MT * 4;  // 8-row groups per 32-column half
        const int rsub = lane >> 3, slot = lane & 7;
        const int rps = ep.rows_per_seq;
        // folded RMSNorm, producer side (mode 2).  Under an ordered split-K only the last part sees the finished h.
        const bool nf_on = FOLD && MODE == 2 && ep.nf_xg && (ep.ksplit <= 1 || (int)blockIdx.y == ep.ksplit - 1);
        float nf_rs[NT];  // this lane group's row sums of h_new^2 over the wave's 32-column blocks, in block order (b0, or b0 + b1)
#pragma unroll
        for (int t = 0; t < NT; ++t) nf_rs[t] = 0.f;
        // Small tile (128x128, the launches of one- and two-song requests): EVERY load of both column halves is requested before
        // anything is stored.  vmcnt retires in order and counts stores, so a half whose loads sit behind the other half's stores
        // waits for those stores to be acknowledged and then for its own round trip: two serial memory latencies per epilogue, and
        // under the ordered split-K the next part waits for all of it (M = 750: 22 k cycles for the pair of parts; ACE355_GEMM_CLK).
        // The 192x256 tile keeps one half in flight (48 more registers there cost more than the round trip, see below).
        // (Round 4, measured and removed: the big tiles requesting half 1's old H rows as soon as half 0's accumulators are staged, i.e.
        //  ahead of half 0's stores.  100 B of scratch - the per-column vectors spill - and the in-pass probe of this epilogue at M = 6000
        //  went from 42.3 k to 51.7 k cycles, the pass from 515.8 to 523.5 ms (ABAB).  The burst is bandwidth bound as it is: 122 MB per
        //  launch in 23.7 us = 5.1 TB/s, beside 5.4 TB/s for the pure-store burst of the patchify GEMM; profiles/r04/r04_gemm_clk_inpass.txt.)
        constexpr bool PRE = (MODE == 2 && MT == 2 && NTW == 2);
        constexpr int NPJ = PRE ? NTW : 1;
        float4 hvp[NPJ][NT], g1p[NPJ], a2p[NPJ], b2p[NPJ], cvp[NPJ], ngAp[NPJ], ngBp[NPJ];
        if constexpr (PRE) {
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = nw0 + j * 32 + slot * 4;
                const float* hp = reinterpret_cast<const float*>(Cv) + n;
                if (ep.ksplit <= 1 || ep.sk_ord) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int m = mw0 + t * 8 + rsub;
                        hvp[j][t] = ldf4_h(hp + (long)(ROWS_FULL ? m : min(m, M - 1)) * ldc);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const int n = nw0 + j * 32 + slot * 4;
                cvp[j] = {0.f, 0.f, 0.f, 0.f};
                if (ep.cvec && (ep.ksplit <= 1 || blockIdx.y == 0)) cvp[j] = ldf4(ep.cvec + n);
                if (ep.g1) {
                    const int seqA = mw0 / rps, last = (M - 1) / rps;
                    g1p[j] = ldf4(ep.g1 + n);
                    a2p[j] = ldf4(ep.g2 + (long)min(seqA, last) * ep.g2_stride + n);
                    b2p[j] = ldf4(ep.g2 + (long)min(seqA + 1, last) * ep.g2_stride + n);
                }
                if (nf_on) { ngAp[j] = ldf4(ep.nf_gA + n); ngBp[j] = ldf4(ep.nf_gB + n); }
            }
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 a;
                    a.x = aq<L16>(acc[i][j], g, 0); a.y = aq<L16>(acc[i][j], g, 1); a.z = aq<L16>(acc[i][j], g, 2); a.w = aq<L16>(acc[i][j], g, 3);
                    // fp32 staged image: 4-column (16-byte) slots
                    *reinterpret_cast<float4*>(stg + stage_off<128>(i * 32 + q_row<L16>(g, lane), q_col<L16>(g, lane) >> 2)) = a;
                }
            const int n = nw0 + j * 32 + slot * 4;
            float* hp = reinterpret_cast<float*>(Cv) + n;
            if constexpr (MODE == 1) {
                float4 b = {0.f, 0.f, 0.f, 0.f};
                if (ep.bias) b = ldf4(ep.bias + n);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int row = t * 8 + rsub;
                    float4 a = *reinterpret_cast<const float4*>(stg + stage_off<128>(row, slot));
                    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
                    const int m = mw0 + row;
                    if (ROWS_FULL || m < M) *reinterpret_cast<float4*>(hp + (long)m * ldc) = a;
                }
            } else {
                float4 g1 = {1.f, 1.f, 1.f, 1.f}, gA = g1, gB = g1, cv = {0.f, 0.f, 0.f, 0.f};
                int remA = 0;
                const bool two_gate = MT * 32 <= rps;  // the wave's rows touch at most two sequences
                // the old H rows (the big, slow loads) are requested FIRST, the per-column gate / constant vectors behind
                // them: one memory round trip per half instead of three (each gate sum used to wait for its own loads
                // before the H loads were even issued)
                float4 hv[NT];
                if constexpr (PRE) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) hv[t] = hvp[j][t];
                    cv = cvp[j];
                    if (ep.g1) {
                        g1 = g1p[j];
                        remA = mw0 - (mw0 / rps) * rps;
                        gA = {g1.x + a2p[j].x, g1.y + a2p[j].y, g1.z + a2p[j].z, g1.w + a2p[j].w};
                        gB = {g1.x + b2p[j].x, g1.y + b2p[j].y, g1.z + b2p[j].z, g1.w + b2p[j].w};
                    }
                } else {
                if (ep.ksplit <= 1 || ep.sk_ord) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int m = mw0 + t * 8 + rsub;
                        hv[t] = ldf4_h(hp + (long)(ROWS_FULL ? m : min(m, M - 1)) * ldc);
                    }
                }
                if (ep.cvec && (ep.ksplit <= 1 || blockIdx.y == 0)) cv = ldf4(ep.cvec + n);  // (the first K part adds the constant term)
                if (ep.g1) {
                    g1 = ldf4(ep.g1 + n);
                    const int seqA = mw0 / rps, last = (M - 1) / rps;
                    remA = mw0 - seqA * rps;
                    const float4 a2 = ldf4(ep.g2 + (long)min(seqA, last) * ep.g2_stride + n);
                    const float4 b2 = ldf4(ep.g2 + (long)min(seqA + 1, last) * ep.g2_stride + n);
                    gA = {g1.x + a2.x, g1.y + a2.y, g1.z + a2.z, g1.w + a2.w};
                    gB = {g1.x + b2.x, g1.y + b2.y, g1.z + b2.z, g1.w + b2.w};
                }
                }
                if (ep.ksplit > 1 && !ep.sk_ord) {
                    // (Requesting the old H values of both column halves before the barrier / staging - one memory round
                    //  trip instead of two - was tried for the read-modify-write variant below: neutral at M = 6000, and the
                    //  larger live range slowed THIS path by 25-50 %; reverted.)
                    // split-K: this workgroup holds a partial sum over its K range; gate * partial is added with fp32 atomics
                    // (no workspace, no reduction pass), 64 consecutive-lane floats = two full 128-byte rows per instruction.
                    // The constant term is added by the first K range only.
                    const bool add_cv = ep.cvec && blockIdx.y == 0;
                    const int col = lane & 31, rhalf = lane >> 5;
                    const int nn = nw0 + j * 32 + col;
                    float g1s = 1.f, gAs = 1.f, gBs = 1.f, cvs = 0.f;
                    if (ep.cvec) cvs = ep.cvec[nn];
                    if (ep.g1) {
                        g1s = ep.g1[nn];
                        const int seqA = mw0 / rps, last = (M - 1) / rps;
                        gAs = g1s + ep.g2[(long)min(seqA, last) * ep.g2_stride + nn];
                        gBs = g1s + ep.g2[(long)min(seqA + 1, last) * ep.g2_stride + nn];
                    }
                    float* hq = reinterpret_cast<float*>(Cv) + nn;
#pragma unroll
                    for (int t = 0; t < MT * 16; ++t) {
                        const int row = t * 2 + rhalf;
                        const int m = mw0 + row;
                        const float a = *reinterpret_cast<const float*>(stg + stage_off<128>(row, col >> 2) + (col & 3) * 4);
                        float gt = (remA + row >= rps) ? gBs : gAs;
                        if (ep.g1 && !two_gate) gt = g1s + ep.g2[(long)(min(m, M - 1) / rps) * ep.g2_stride + nn];
                        float o = gt * a;
                        if (add_cv && m >= ep.cvec_row0) o += cvs;
                        if (ROWS_FULL || m < M) unsafeAtomicAdd(hq + (long)m * ldc, o);
                    }
                    continue;
                }
                // The per-row gate lookup of short sequences is a separate loop: as a branch inside the store loop its (skipped)
                // load still left an `s_waitcnt vmcnt(0)` at the join, and vmcnt counts stores too - every 16-byte store of the
                // residual update waited for the previous one's acknowledgement (25-28 k cycles per tile).
                float4 ngA = {0.f, 0.f, 0.f, 0.f}, ngB = ngA;
                if constexpr (PRE) {
                    if (nf_on) { ngA = ngAp[j]; ngB = ngBp[j]; }
                } else {
                    if (nf_on) { ngA = ldf4(ep.nf_gA + n); ngB = ldf4(ep.nf_gB + n); }
                }
                // the next norm's operand leaves with the row: xg = bf16(h_new * g) (8 lanes x 8 bytes = a 64-byte half line per row),
                // the sum of h_new^2 over these 32 columns = the 8 lanes of the row adds up in nf_rs.  (Keeping the packed values until
                // both column halves are done and writing whole 128-byte rows through the staging image costs 48 registers: the 192x256
                // kernel then spills; the stores themselves measured 2.3 us per launch as they are.)
                auto nf_emit = [&](int t, int m, const float4& o) {
                    const float4 gq = (m < ep.nf_split) ? ngA : ngB;
                    uint2 pk;
                    pk.x = pack_bf2(o.x * gq.x, o.y * gq.y);
                    pk.y = pack_bf2(o.z * gq.z, o.w * gq.w);
                    if (ROWS_FULL || m < M) stu2_xg(ep.nf_xg + (long)m * ep.nf_ldx + n, pk);
                    float sq = o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
                    sq = dpp_add<0xB1>(sq);    // quad_perm [1,0,3,2]
                    sq = dpp_add<0x4E>(sq);    // quad_perm [2,3,0,1]
                    sq = dpp_add<0x141>(sq);   // row_half_mirror: the other quad of the 8 lanes
                    nf_rs[t] += sq;
                };
                if (!ep.g1 || two_gate) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int row = t * 8 + rsub;
                        const int m = mw0 + row;
                        const float4 a = *reinterpret_cast<const float4*>(stg + stage_off<128>(row, slot));
                        const float4 gt = (remA + row >= rps) ? gB : gA;
                        float4 o = {hv[t].x + gt.x * a.x, hv[t].y + gt.y * a.y, hv[t].z + gt.z * a.z, hv[t].w + gt.w * a.w};
                        if (ep.cvec && m >= ep.cvec_row0) { o.x += cv.x; o.y += cv.y; o.z += cv.z; o.w += cv.w; }
                        if (ROWS_FULL || m < M) stf4_h(hp + (long)m * ldc, o);
                        if (nf_on) nf_emit(t, m, o);
                    }
                } else {  // short sequences (tiny configs): a wave's rows touch more than two sequences
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int row = t * 8 + rsub;
                        const int m = mw0 + row;
                        const float4 a = *reinterpret_cast<const float4*>(stg + stage_off<128>(row, slot));
                        const float4 r2 = ldf4(ep.g2 + (long)(min(m, M - 1) / rps) * ep.g2_stride + n);
                        const float4 gt = {g1.x + r2.x, g1.y + r2.y, g1.z + r2.z, g1.w + r2.w};
                        float4 o = {hv[t].x + gt.x * a.x, hv[t].y + gt.y * a.y, hv[t].z + gt.z * a.z, hv[t].w + gt.w * a.w};
                        if (ep.cvec && m >= ep.cvec_row0) { o.x += cv.x; o.y += cv.y; o.z += cv.z; o.w += cv.w; }
                        if (ROWS_FULL || m < M) stf4_h(hp + (long)m * ldc, o);
                        if (nf_on) nf_emit(t, m, o);
                    }
                }
            }
        }
        if constexpr (MODE == 2 && FOLD) {
            if (nf_on) {   // (workgroup-uniform)
                // A row's sum of squares leaves the workgroup as ONE integer per aligned 128-COLUMN GROUP, its four 32-column blocks (each the fixed 8-lane
                // sum above) added as (b0 + b1) + (b2 + b3) in fp32, then 2^-24 fixed point, then a memory-side integer atomic.  Integer adds commute, and
                // the fp32 part is the same expression in every tile form: a wave with two blocks holds b_even + b_odd already (the 128- and 256-wide tiles:
                // the pairs meet through LDS), four one-block waves (the 8-wave 192 x 128 tile) pair up here.
                // The folded norm's rstd therefore no longer depends on the tile a row happened to be computed in (round 6; until then the partials were
                // summed wave after wave across the whole tile width, and with them the last bit of rstd followed the launch shape).
                if (slot == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) xw[wave * (MT * 32) + t * 8 + rsub] = nf_rs[t];
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_s_barrier();
                // (one atomic per row and TILE as before: a 256-wide tile converts its two groups separately and adds the integers)
                if (wave % wnw == 0 && slot == 0) {
                    auto fix = [](float v) { return (unsigned long long)(long long)fminf(v * 16777216.f, 1.4e17f); };   // 2^-24 units; capped at 2^57: the partials of a row cannot wrap
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float* x = xw + wave * (MT * 32) + t * 8 + rsub;
                        unsigned long long q;
                        if constexpr (NTW == 2) {
                            q = fix(x[0] + x[MT * 32]);
                            if (wnw == 4) q += fix(x[2 * MT * 32] + x[3 * MT * 32]);
                        } else {
                            q = fix((x[0] + x[MT * 32]) + (x[2 * MT * 32] + x[3 * MT * 32]));
                        }
                        const int m = mw0 + t * 8 + rsub;
                        if (ROWS_FULL || m < M) atomicAdd((m < ep.nf_split ? ep.nf_sqA : ep.nf_sqB) + m, q);
                    }
                }
            }
        }
    }
}

// Per-element path for tiles that are partial in N or whose C / vectors are not 16-byte aligned (unit-test shapes, tiny configs).
template <int MODE, int MT, int NTW, bool L16 = false>
__device__ __forceinline__ void gemm_epilogue_scalar(AccTile<L16> (&acc)[MT][NTW], void* __restrict__ Cv, int ldc, int M, int N,
                                                     const GemmEpilogue& ep, int mw0, int nw0, int lane) {
    constexpr int J1 = NTW - 1;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int m = mw0 + i * 32 + q_row<L16>(g, lane);
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < (MODE == 3 ? 1 : NTW); ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int nl = j * 32 + q_col<L16>(g, lane) + e;
                    if (MODE == 3) {
                        if (nw0 + 32 + nl < N) reinterpret_cast<bf16_t*>(Cv)[(long)m * ldc + (nw0 >> 1) + nl] = f2bf(silu_f(aq<L16>(acc[i][0], g, e)) * aq<L16>(acc[i][J1], g, e));
                        continue;
                    }
                    const int n = nw0 + nl;
                    if (n >= N) continue;
                    const float v = aq<L16>(acc[i][j], g, e);
                    if (MODE == 0 || MODE == 4) reinterpret_cast<bf16_t*>(Cv)[(long)m * ldc + n] = f2bf(v + (ep.bias ? ep.bias[n] : 0.f));  // (4: launch_gemm never sends the head epilogue here)
                    else if (MODE == 1) reinterpret_cast<float*>(Cv)[(long)m * ldc + n] = v + (ep.bias ? ep.bias[n] : 0.f);
                    else {
                        float gate = 1.f;
                        if (ep.g1) gate = ep.g1[n] + ep.g2[(long)(m / ep.rows_per_seq) * ep.g2_stride + n];
                        float add = gate * v;
                        if (ep.cvec && m >= ep.cvec_row0 && (ep.ksplit <= 1 || blockIdx.y == 0)) add += ep.cvec[n];
                        if (ep.ksplit > 1 && !ep.sk_ord) unsafeAtomicAdd(reinterpret_cast<float*>(Cv) + (long)m * ldc + n, add);
                        else reinterpret_cast<float*>(Cv)[(long)m * ldc + n] += add;
                    }
                }
        }
}

// smem: the workgroup's LDS (dead after the K loop; every wave stages MT*32 rows x 128 B in its own slice of it).
template <int MODE, int MT, int NTW = 2, bool FOLD = true, bool L16 = false>
__device__ __forceinline__ void gemm_epilogue(AccTile<L16> (&acc)[MT][NTW], char* smem, void* __restrict__ Cv, int ldc, int M, int N,
                                              const GemmEpilogue& ep, int m0, int n0, int wm, int wn, int wave, int lane,
                                              int bm = MT * 64, int bn = 128, const char* vec = nullptr) {
    const int mw0 = m0 + wm * (MT * 32), nw0 = n0 + wn * (NTW * 32);
    if (ep.wide_ok && n0 + bn <= N) {  // workgroup-uniform
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();  // every wave is done reading operand fragments: the stages may be overwritten
        char* stg = smem + wave * (MT * 32 * 128);
        float* xw = reinterpret_cast<float*>(smem + (bn / (NTW * 32)) * (bm / (MT * 32)) * (MT * 32 * 128));  // behind the last staging slice
        const int wnw = bn / (NTW * 32);
        const int mt0 = wm * (MT * 32), nt0 = wn * (NTW * 32);
        if (m0 + bm <= M) gemm_epilogue_wide<MODE, MT, NTW, true, FOLD, L16>(acc, stg, Cv, ldc, M, ep, mw0, nw0, lane, xw, wave, wnw, vec, mt0, nt0);
        else gemm_epilogue_wide<MODE, MT, NTW, false, FOLD, L16>(acc, stg, Cv, ldc, M, ep, mw0, nw0, lane, xw, wave, wnw, vec, mt0, nt0);
    } else {
        gemm_epilogue_scalar<MODE, MT, NTW, L16>(acc, Cv, ldc, M, N, ep, mw0, nw0, lane);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ W,
                                                       int ldw, void* __restrict__ Cv, int ldc, int M, int N, int K,
                                                       GemmEpilogue ep, int tiles_n, int nwg) {
    __shared__ __attribute__((aligned(16))) char smem[65536];  // 2 stages x (A 16 KB | W 16 KB)

    // XCD-aware bijective remap: block b runs on XCD b % 8; give XCD x a contiguous tile range.
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // global -> register staging assignment: 4 x 16-B chunks of A and of W per thread per K-step.
    // chunk c = tid + i*256 -> row = (tid>>3) + 32*i, slot = tid&7; the swizzle term (row>>1)&7 is i-invariant.
    const int srow = tid >> 3, sslot = tid & 7;
    const int st_off0 = lds_off(srow, sslot);
    const bf16_t* a_src0 = A + (long)min(m0 + srow, M - 1) * lda + sslot * 8;
    const bf16_t* a_src1 = A + (long)min(m0 + srow + 32, M - 1) * lda + sslot * 8;
    const bf16_t* a_src2 = A + (long)min(m0 + srow + 64, M - 1) * lda + sslot * 8;
    const bf16_t* a_src3 = A + (long)min(m0 + srow + 96, M - 1) * lda + sslot * 8;
    const bf16_t* w_src0 = W + (long)min(n0 + srow, N - 1) * ldw + sslot * 8;
    const bf16_t* w_src1 = W + (long)min(n0 + srow + 32, N - 1) * ldw + sslot * 8;
    const bf16_t* w_src2 = W + (long)min(n0 + srow + 64, N - 1) * ldw + sslot * 8;
    const bf16_t* w_src3 = W + (long)min(n0 + srow + 96, N - 1) * ldw + sslot * 8;
#define LOAD_TILE(koff)                                            \
    ra0 = *reinterpret_cast<const uint4*>(a_src0 + (koff));        \
    ra1 = *reinterpret_cast<const uint4*>(a_src1 + (koff));        \
    ra2 = *reinterpret_cast<const uint4*>(a_src2 + (koff));        \
    ra3 = *reinterpret_cast<const uint4*>(a_src3 + (koff));        \
    rw0 = *reinterpret_cast<const uint4*>(w_src0 + (koff));        \
    rw1 = *reinterpret_cast<const uint4*>(w_src1 + (koff));        \
    rw2 = *reinterpret_cast<const uint4*>(w_src2 + (koff));        \
    rw3 = *reinterpret_cast<const uint4*>(w_src3 + (koff));
#define STORE_TILE(base)                                                    \
    *reinterpret_cast<uint4*>((base) + st_off0) = ra0;                      \
    *reinterpret_cast<uint4*>((base) + st_off0 + 4096) = ra1;               \
    *reinterpret_cast<uint4*>((base) + st_off0 + 8192) = ra2;               \
    *reinterpret_cast<uint4*>((base) + st_off0 + 12288) = ra3;              \
    *reinterpret_cast<uint4*>((base) + 16384 + st_off0) = rw0;              \
    *reinterpret_cast<uint4*>((base) + 16384 + st_off0 + 4096) = rw1;       \
    *reinterpret_cast<uint4*>((base) + 16384 + st_off0 + 8192) = rw2;       \
    *reinterpret_cast<uint4*>((base) + 16384 + st_off0 + 12288) = rw3;

    AccTile<false> acc[2][2];   // (the bring-up kernel keeps the 32x32x16 MFMA)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc_zero<false>(acc[i][j]);

    uint4 ra0, ra1, ra2, ra3, rw0, rw1, rw2, rw3;
    LOAD_TILE(0)
    STORE_TILE(smem)
    __syncthreads();

    const int nk = K / BK;
    const int frow = lane & 31, fhalf = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1) < nk;
        if (more) { LOAD_TILE((kt + 1) * BK) }
        const char* As = smem + (kt & 1) * 32768;
        const char* Ws = As + 16384;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 fa[2], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = as_bf16x8(*reinterpret_cast<const uint4*>(As + lds_off(wm * 64 + i * 32 + frow, kk * 2 + fhalf)));
                fw[i] = as_bf16x8(*reinterpret_cast<const uint4*>(Ws + lds_off(wn * 64 + i * 32 + frow, kk * 2 + fhalf)));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j].v = mfma32(fw[j], fa[i], acc[i][j].v);
        }
        if (more) {
            char* Ad = smem + ((kt + 1) & 1) * 32768;
            STORE_TILE(Ad)
        }
        __syncthreads();
    }

    gemm_epilogue<MODE, 2, 2, true, false>(acc, smem, Cv, ldc, M, N, ep, m0, n0, wm, wn, wave, lane);
}


// ------------------------------------------------------------------------------------------------ LDS-DMA helpers
// hipcc treats the global_load_lds builtin as a store to LDS that may alias the fragment reads and drains it
// (s_waitcnt vmcnt(0)) before the first ds_read of the same iteration: v2 never overlaps a tile's DMA with MFMAs inside
// a workgroup.  v3 issues the DMA from inline asm (invisible to the compiler's dependence tracking), keeps NS LDS stages
// with NS-1 tiles in flight, waits with a COUNTED vmcnt for exactly the tile about to be consumed, and uses the raw
// s_barrier (no implicit drain).  Per K-step:  wait(tile kt) -> barrier -> issue(tile kt+NS-1) -> MFMAs(tile kt).
//   RAW: a wave's vmcnt covers its own DMA pieces; the barrier extends that to every wave's pieces.
//   WAR: tile kt+NS-1 lands in the stage read at kt-1; every wave passed this iteration's barrier after finishing kt-1.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}
// same, address = uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset: the per-lane part is loop invariant, so a
// K step costs scalar adds only (the flat form needs a 64-bit VALU add per piece)
__device__ __forceinline__ void glds16_sv(unsigned voff, const void* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
// One MX-scaled fp8 MFMA over 64 K elements: (x0 | x1) = the two 16-byte fragments a lane holds for one operand (slots 2kk, 2kk+1),
// sa / sb = the lane's scale words, OPSEL = the byte of them that belongs to this MFMA's k-block (tools/probe/mx_probe.py).
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
template <int OPSEL>
__device__ __forceinline__ f32x16 mfma_mx(bf16x8 a0, bf16x8 a1, bf16x8 b0, bf16x8 b1, f32x16 c, unsigned sa, unsigned sb) {
    const i32x4_t A0 = __builtin_bit_cast(i32x4_t, a0), A1 = __builtin_bit_cast(i32x4_t, a1);
    const i32x4_t B0 = __builtin_bit_cast(i32x4_t, b0), B1 = __builtin_bit_cast(i32x4_t, b1);
    const i32x8_t A = {A0[0], A0[1], A0[2], A0[3], A1[0], A1[1], A1[2], A1[3]};
    const i32x8_t B = {B0[0], B0[1], B0[2], B0[3], B1[0], B1[1], B1[2], B1[3]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 0, 0, OPSEL, (int)sa, OPSEL, (int)sb);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------ v4: rotated pipeline
// v3 still parks the matrix pipe at every K-step boundary: after the barrier a wave must issue its DMA pieces and wait
// a full LDS round trip for the first fragments before any MFMA can start.  v4 rotates the loop by half a K-step so the
// barrier sits BETWEEN two MFMA groups whose operands are already in registers:
//     ds_read frags(kt, kk=2,3) -> Q   |  MFMA(kt, kk=0,1) from P            (P was loaded before the previous barrier)
//     lgkmcnt(0); vmcnt(0) [tile kt+1]; s_barrier                            (stage kt&1 is now dead for every wave)
//     DMA tile kt+2 -> stage kt&1;  ds_read frags(kt+1, kk=0,1) -> P  |  MFMA(kt, kk=2,3) from Q
// Two LDS stages (2 workgroups per CU), one barrier per K-step, >= 8 MFMAs of cover on both sides of every LDS read.
// WNW = waves along N (2 -> BN 128, 256 threads, 2 workgroups/CU; 4 -> BN 256, 512 threads, 1 workgroup/CU).
// NTW = 32-column accumulator tiles per wave (2: wave tile MT*32 x 64; 1: MT*32 x 32, used for the 8-wave 192x128 tile).
// PERS 1: persistent workgroups - the grid is one workgroup per CU slot and each walks its XCD region's tiles with stride
// gridDim/8: no s_endpgm store drain, no workgroup re-dispatch and no kernarg reload between the tiles of a multi-round GEMM.
// NS = LDS stages (2: product default, one K step of DMA cover; 3-4: small-M launches with one workgroup per CU, where
// nothing else hides the HBM latency of the weight stream: NS-1 tiles stay in flight behind a counted vmcnt).
// FP8 = 1: MX-scaled fp8 operands (v_mfma_scale_f32_32x32x64_f8f6f4).  A K step is 128 fp8 elements = the same 128 bytes per row,
// so the LDS images, DMA pieces and fragment reads are byte-identical to the bf16 kernel; the two bf16 MFMAs over the fragments
// of slots (2 kk, 2 kk + 1) become ONE scaled MFMA over both (lane half h holds k = 16h..16h+15 and 32+16h..+15 of the 64: exactly
// those two 16-byte slots; tools/probe/mx_probe.py established the layout).  Scales ride as one extra 1 KB DMA piece per operand per
// K step (waves 0 / 1) behind the tiles: word [row] = the four E8M0 bytes of this K step; the upper lane half shifts its word by 8
// so that op_sel 0 / 2 picks block (0 | 1) / (2 | 3).
// Round 4, built / measured / removed (git history: "Intra-workgroup split-K"): KG = 2, two wave groups per workgroup, each running this
// pipeline on its own two LDS stages over half of the workgroup's K range, accumulators exchanged through LDS at the end (64 KB, no L2
// round trip) - meant for the small-M launches, where a 128x128 workgroup is alone on its CU with one wave per SIMD.  Correct (100 GPU
// tests), no scratch once the residual epilogue's prefetch registers were dropped for it, and much slower: 181.8 vs 147.4 ms per batch-1
// request (configs[0]: 64.1 vs 48.3 ms).  The K loop of those launches is bound by what ONE CU can pull through the L2 -> LDS DMA path
// (32 KB per K step at ~70 GB/s = 0.46 us, the in-pass figure of DESIGN.md section 10), not by MFMA issue: a second wave group adds no
// ingest bandwidth, and the two-stage rings it leaves room for (2 x 2 x 32 KB) hide less latency than this kernel's four stages.
// GemmEpilogue::pf_*: workgroup `rank` of the pf_x prefetch workgroups of XCD `xcd` reads its share of the W rows the next launch's tiles on this
// XCD will stream (column region xj of that launch's XCD grid) and discards them: plain 16-byte loads, eight in flight per lane, which leave
// the lines in this XCD's L2.
__device__ __noinline__ void gemm_prefetch_region(const void* pf_w, unsigned pf_chunk16, unsigned pf_total16, unsigned pf_len16, int pf_xcd_n, int pf_x, int xcd, int rank, int nthr) {
    typedef unsigned int u32x4p __attribute__((ext_vector_type(4)));
    const int xi = xcd / pf_xcd_n, xj = xcd - xi * pf_xcd_n;
    const unsigned long long base = (unsigned long long)xj * pf_chunk16;
    if (base >= pf_total16) return;
    const unsigned long long n = min((unsigned long long)pf_len16, (unsigned long long)pf_total16 - base);
    const u32x4p* p = reinterpret_cast<const u32x4p*>(pf_w) + base;
    const unsigned long long nt = (unsigned long long)pf_x * nthr;
    unsigned long long i = (unsigned long long)rank * nthr + threadIdx.x;
    unsigned acc = 0;
    for (; i + 7 * nt < n; i += 8 * nt) {
        u32x4p v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * nt];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].w;
    }
    for (; i < n; i += nt) acc ^= p[i].x;
    if (acc == 0x9e3779b9u && rank == 0x7fffffff) asm volatile("s_nop 0" ::"v"(acc));   // (keeps the loads; never taken)
}

template <int MODE, int MT, int WNW, int NTW = 2, int PERS = 0, int NS = 2, int FP8 = 0>
__global__ __launch_bounds__(WNW * 128, 2) void gemm_sp_kernel(const bf16_t* __restrict__ A0, int lda, const bf16_t* __restrict__ W0,
                                                          int ldw, void* __restrict__ Cv, int ldc, int M, int N, int K,
                                                          GemmEpilogue ep, int tiles_n, int nwg, int group_m, int xcd_m) {
    constexpr int BMv = MT * 64;
    constexpr int A_BYTES = BMv * 128;
    constexpr int BNv = WNW * NTW * 32;
    static_assert(MODE != 3 || NTW == 2, "SwiGLU pairs two column tiles per wave");
    constexpr int NW = 2 * WNW;                 // waves per workgroup
    constexpr int W_BYTES = BNv * 128;
    constexpr int SC_BYTES = FP8 ? 2048 : 0;    // MX scales of a K step: 1 KB (256 rows x 4 B) per operand
    constexpr int STAGE = A_BYTES + W_BYTES + SC_BYTES;
    static_assert(!FP8 || (NS == 2 && BMv <= 256 && WNW * NTW * 32 <= 256 && 2 * WNW >= 2), "MX path: 2 stages, tiles up to 256 rows");
    constexpr int AJ = BMv / (8 * NW);          // A DMA pieces per wave per tile (8 rows each)
    constexpr int WJ = BNv / (8 * NW);          // W DMA pieces per wave per tile
    static_assert(BMv % (8 * NW) == 0 && BNv % (8 * NW) == 0, "tile rows must split evenly over the waves");
    constexpr int EPI_BYTES = NW * MT * 32 * 128 + NW * NTW * MT * 32 * 4;  // epilogue staging: MT*32 rows x 128 B per wave (+ mode 4's row sums per 32-column block)
    // Vector area (8-wave bf16 kernels whose epilogue consumes a folded norm; one workgroup per CU, so the 4 KB are free): see gemm_epilogue_wide
    constexpr bool VEC = !FP8 && WNW == 4 && (MODE == 0 || MODE == 3 || MODE == 4);
    constexpr int VOFF = NS * STAGE > EPI_BYTES ? NS * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[VOFF + (VEC ? 4096 : 0)];

    // Workgroup -> tile map.  Block b runs on XCD b % 8 (observed, speed only).  The 8 XCDs (private L2s) form an
    // xcd_m x xcd_n grid of rectangular tile regions, chosen per GEMM to minimise the bytes each L2 must pull from
    // HBM/MALL (xcd_n * |A| + xcd_m * |W|); inside a region tiles are walked in groups of `group_m` rows so the ~32
    // workgroups resident on one XCD share both A row-panels and W column-panels.
    for (int idx = blockIdx.x >> 3;; idx += (int)(gridDim.x >> 3)) {
    const unsigned long long t_entry = ep.clk_probe ? clock64() : 0ull;
    int tm, tn, trow;   // trow: the tile's row inside its XCD region
    {
        const int tiles_m = nwg / tiles_n;
        const int xcd = blockIdx.x & 7;
        const int xcd_n = 8 / xcd_m;
        const int rm = (tiles_m + xcd_m - 1) / xcd_m, rn = (tiles_n + xcd_n - 1) / xcd_n;  // region size in tiles
        if constexpr (PERS == 0 && !FP8) {
            // the workgroups behind the region's rm * rn slots (GemmEpilogue::pf_x per XCD) prefetch the NEXT launch's weight rows of this XCD
            if (ep.pf_x > 0 && idx >= rm * rn) {
                if (blockIdx.y == 0) gemm_prefetch_region(ep.pf_w, ep.pf_chunk16, ep.pf_total16, ep.pf_len16, ep.pf_xcd_n, ep.pf_x, xcd, idx - rm * rn, (int)blockDim.x);
                return;
            }
        }
        const int xi = xcd / xcd_n, xj = xcd - xi * xcd_n;
        const int m_lo = xi * rm, n_lo = xj * rn;
        const int hm = min(rm, tiles_m - m_lo), hn = min(rn, tiles_n - n_lo);  // this region's extent (may be ragged / empty)
        if (hm <= 0 || hn <= 0 || idx >= hm * hn) return;
        const int gsz = group_m * hn;
        const int grp = idx / gsz, rem = idx - grp * gsz;
        const int first_m = grp * group_m;
        const int gm = min(hm - first_m, group_m);
        trow = first_m + rem % gm;
        tm = m_lo + trow;
        tn = n_lo + rem / gm;
    }
    const int m0 = tm * BMv, n0 = tn * BNv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (waves are dealt round-robin over the SIMDs: in the 8-wave tiles waves w and w + 4 share one, i.e. the same W columns and different A rows)
    const int wm = wave / WNW, wn = wave % WNW;
    const int vw = wm * WNW + wn;

    const int lrow = lane >> 3, pslot = lane & 7;
    const int sslot = pslot ^ ((4 * (wave & 1) + (lrow >> 1)) & 7);
    unsigned a_voff[AJ], w_voff[WJ];  // per-lane byte offsets from A / W (launch_gemm checks both matrices are < 4 GB)
    // ACE355_GEMM_CLK bits 1 / 2 (diagnostic, WRONG results): every workgroup streams the A / W panel of tile 0 - the operands come out of
    // the L2 whatever the tile, which separates what the fabric costs a K step from what the CU-side path (DMA, LDS, MFMA) costs it
    const int lm0 = (ep.clk_probe & 2) ? 0 : m0, ln0 = (ep.clk_probe & 4) ? 0 : n0;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int ar = lm0 + 8 * (wave + NW * j) + lrow;
        const int arw = (ep.a_wrap > 0 && ar >= ep.a_wrap) ? ar - ep.a_wrap : ar;   // (second half of the rows = the first half's operand)
        a_voff[j] = ((unsigned)(ar < M ? arw : (ep.a_zero_idx > 0 ? ep.a_zero_idx : M - 1 - (ep.a_wrap > 0 ? ep.a_wrap : 0))) * (unsigned)lda + sslot * 8) * 2u;   // (pad rows: the zero row, else the last row)
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) w_voff[j] = ((unsigned)min(ln0 + 8 * (wave + NW * j) + lrow, N - 1) * (unsigned)ldw + sslot * 8) * 2u;
    const unsigned lds0 = (unsigned)(uintptr_t)smem + (unsigned)wave * 1024u;

    constexpr bool L16 = !FP8;   // bf16: v_mfma_f32_16x16x32_bf16 accumulators (AccTile<true>); MX fp8: the scaled 32x32x64 form
    AccTile<L16> acc[MT][NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc_zero<L16>(acc[i][j]);

    // split-K (ep.kparts > 1): blockIdx.y owns K steps [kt0, kt0 + nk)
    const int nk_all = K / BK;
    const int kchunk = (nk_all + ep.kparts - 1) / ep.kparts;
    const int kt0 = (ep.kparts > 1) ? (int)blockIdx.y * kchunk : 0;
    const int nk = (ep.kparts > 1) ? min(kchunk, nk_all - kt0) : nk_all;
    if (nk <= 0) return;
    const bf16_t* A = A0 + kt0 * BK;
    const bf16_t* W = W0 + kt0 * BK;
    // K rotation (ep.krot; one-round launches): the workgroups of XCD x start at K step x * nk / 8 and wrap, so the eight L2s do not miss the
    // same K slice at the same moment: a slice's first reader pays the HBM latency, the other seven find it in the memory-side cache
    // (launches that leave a quarter of the CUs without a tile have no common miss to spread: batch 1 measured + 0.3 % with the rotation on)
    // Row stagger (ep.krot >> 4 = d > 0): inside an XCD the workgroups of region row r start another r * d K steps ahead.  All 32 workgroups
    // of an XCD used to reach every new K slice together - 32 "first readers" waiting for the same fabric fetch - where in the persistent
    // launches they drift apart and all but one find the slice in their L2; the workgroups of one region row (which share the A slice) stay
    // in step, the rows that share a W slice follow each other d steps apart (a few 100 KB of traffic later: still in the 4 MB L2).
    const int krot = (!PERS && !FP8 && ep.krot && ep.kparts == 1 && nwg >= 192) ? ((int)(blockIdx.x & 7) * (nk >> 3) + (ep.krot >> 4) * trow) % nk : 0;
    auto kmap = [&](int kt) -> int { const int k = kt + krot; return k >= nk ? k - nk : k; };
    const int frow = L16 ? (lane & 15) : (lane & 31), fhalf = L16 ? (lane >> 4) : (lane >> 5);   // fragment row inside its 16- / 32-row group, K group of the lane
    // MX scales: wave 0 stages the A rows' words of K step kt (rows m0 .. m0+255 of scale row kt, 16 bytes per lane), wave 1 the W rows'
    const unsigned sc_voff = (unsigned)lane * 16u;
    const unsigned sc_lds = (unsigned)(uintptr_t)smem + A_BYTES + W_BYTES + (wave == 1 ? 1024u : 0u);
    auto issue_scales = [&](int kt) {
        if constexpr (FP8) {
            if (wave < 2) {
                const uint32_t* src = wave == 0 ? ep.mx_sa + (long)(kt0 + kt) * ep.mx_sa_ld + m0 : ep.mx_sw + (long)(kt0 + kt) * ep.mx_sw_ld + n0;
                glds16_sv(sc_voff, src, (unsigned)__builtin_amdgcn_readfirstlane((int)(sc_lds + (unsigned)(kt % NS) * STAGE)));
            }
        }
    };
    auto issue = [&](int kt) {
        const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kt % NS) * STAGE);
#pragma unroll
        for (int j = 0; j < AJ; ++j) glds16_sv(a_voff[j], A + kmap(kt) * BK, sb + j * (NW * 1024));
#pragma unroll
        for (int j = 0; j < WJ; ++j) glds16_sv(w_voff[j], W + kmap(kt) * BK, sb + A_BYTES + j * (NW * 1024));
        issue_scales(kt);
    };
    // this lane's scale words of a stage (row of each of its A / W tiles), pre-shifted for the upper lane half
    unsigned sca[MT], scw[NTW];
    auto load_scales = [&](const char* st) {
        if constexpr (FP8) {
            const char* sc = st + A_BYTES + W_BYTES;
#pragma unroll
            for (int i = 0; i < MT; ++i) sca[i] = *reinterpret_cast<const unsigned*>(sc + (wm * (MT * 32) + i * 32 + frow) * 4) >> (8 * fhalf);
#pragma unroll
            for (int j = 0; j < NTW; ++j) scw[j] = *reinterpret_cast<const unsigned*>(sc + 1024 + (wn * (NTW * 32) + j * 32 + frow) * 4) >> (8 * fhalf);
        }
    };
    // per-lane fragment byte offsets inside a stage for kk = 0 (the kk term only flips slot bits: see lds_off)
    int a_off[MT], w_off[NTW];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_off[i] = (wm * (MT * 32) + i * 32 + frow) * 128;
#pragma unroll
    for (int j = 0; j < NTW; ++j) w_off[j] = A_BYTES + (wn * (NTW * 32) + j * 32 + frow) * 128;
    const int swz = ((wm * (MT * 32) + frow) >> 1) & 7;   // (row>>1)&7 is the same for every 32-row tile of this lane
    const int swzw = ((wn * (NTW * 32) + frow) >> 1) & 7;
    // Fragment (h, tile) of a half K step (kk0 = 0: k 0..31 of the step, kk0 = 2: k 32..63):
    //   32x32x16 form: h = the 16-wide k sub-step, 32-row fragments, slot = (kk0 + h) * 2 + (lane >> 5)
    //   16x16x32 form: h = the 16-ROW half of the 32-row block, the fragment spans the half step's 32 k: slot = (kk0 / 2) * 4 + (lane >> 4)
    // (the same LDS image and the same XOR swizzle serve both: (row >> 1) & 7 does not change across 16-row halves, and the 16 lanes a
    //  ds_read_b128 services together still fall on 16 different 16-byte bank groups)
    auto frag_slot = [&](int kk0, int h) -> int { return L16 ? (kk0 >> 1) * 4 + fhalf : (kk0 + h) * 2 + fhalf; };
    constexpr int HROW = L16 ? 16 * 128 : 0;   // byte offset of fragment half h inside its block (16x16x32 form)
    auto load_frags = [&](const char* st, int kk0, bf16x8 (&fa)[2][MT], bf16x8 (&fw)[2][NTW]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int slot = frag_slot(kk0, h);
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[h][i] = as_bf16x8(*reinterpret_cast<const uint4*>(st + a_off[i] + h * HROW + ((slot ^ swz) << 4)));
#pragma unroll
            for (int j = 0; j < NTW; ++j) fw[h][j] = as_bf16x8(*reinterpret_cast<const uint4*>(st + w_off[j] + h * HROW + ((slot ^ swzw) << 4)));
        }
    };

    bf16x8 pa[2][MT], pw[2][NTW], qa[2][MT], qw[2][NTW];
    // Epilogue vectors of this tile -> LDS vector area, one DMA piece per wave 0-3 BEHIND the prologue's operand tiles (ahead of them
    // they held tile 0 back: loads retire in order and the row sums are a fabric round trip - prologue + 500 cycles, the epilogues' gain
    // gone; profiles/r05/r05_epi_vec_first_ab.txt).  As the youngest request a piece only makes the counted waits of its wave conservative by
    // one, the first K step's wait covers it and that step's barrier publishes it.  Waves 0 / 1: the row sums of squares (two rows per
    // lane; M even, so a pair never straddles the end of the array: rows >= M take the last pair's values and are never stored), wave 2
    // the bias of the tile's columns, wave 3 the head's norm weights.  Only whole tiles on the wide path (workgroup-uniform).
    bool vec_on = false;
    auto vec_issue = [&]() {
    if constexpr (VEC) {
        vec_on = ep.nc_rowsq && ep.wide_ok && (M & 1) == 0 && n0 + BNv <= N && ((uintptr_t)ep.nc_rowsq & 15) == 0 &&
                 (MODE != 4 || (((uintptr_t)ep.hn_wq | (uintptr_t)ep.hn_wk) & 15) == 0);
        if (vec_on) {
            const unsigned vl = (unsigned)(uintptr_t)smem + VOFF;
            if (wave < 2) {
                const unsigned ro = (unsigned)min(m0 + wave * 128 + 2 * lane, M - 2) * 8u;
                glds16_sv(ro, ep.nc_rowsq, (unsigned)__builtin_amdgcn_readfirstlane((int)(vl + (unsigned)wave * 1024u)));
            } else if (wave == 2) {
                if (ep.nc_bias) glds16_sv((unsigned)min(n0 + 4 * lane, N - 4) * 4u, ep.nc_bias, (unsigned)__builtin_amdgcn_readfirstlane((int)(vl + 2048u)));
            } else if (wave == 3) {
                if constexpr (MODE == 4) {
                    if (n0 < ep.hn_qk_cols) glds16_sv((unsigned)((4 * lane) & 127) * 4u, n0 < ep.hn_q_cols ? ep.hn_wq : ep.hn_wk, (unsigned)__builtin_amdgcn_readfirstlane((int)(vl + 3072u)));
                }
            }
        }
    }
    };
    issue(0);
    if (NS == 2) {
        if (nk > 1) issue(1);
        vec_issue();
        if (nk > 1) {
            if (FP8 && wave < 2) wait_vmcnt<AJ + WJ + 1>(); else wait_vmcnt<AJ + WJ>();  // (waves 0 / 1 carry one scale piece per K step)
        } else wait_vmcnt<0>();
    } else {
        // deep pipeline: all NS stages primed when the K range is long enough (else plain drain: short ranges are rare here)
        if (nk >= NS) {
#pragma unroll
            for (int t = 1; t < NS; ++t) issue(t);
            vec_issue();
            wait_vmcnt<(NS - 1) * (AJ + WJ)>();
        } else {
            for (int t = 1; t < nk; ++t) issue(t);
            vec_issue();
            wait_vmcnt<0>();
        }
    }
    __builtin_amdgcn_s_barrier();
    load_frags(smem, 0, pa, pw);
    constexpr bool PAIRK = L16 && NTW == 2 && NW == 8 && MT == 3 && (AJ + WJ) == 7;   // (kstep_pair below)
    constexpr bool PAIRK1 = L16 && NTW == 1 && NW == 8 && MT == 3 && (AJ + WJ) == 5;   // (kstep_pair1: the 192x128 tile)
    if constexpr (PAIRK || PAIRK1) load_frags(smem, 2, qa, qw);   // the pair loop starts a K step with both K halves of its A rows and of column block 0 in registers
    load_scales(smem);

    // ---- ILV: explicit instruction interleave.  An LDS-DMA piece costs 60-185 cycles to ISSUE (TA queue); seven of them
    // back to back right after the barrier stall the in-order wave before its first MFMA.  Here every MFMA is followed
    // by at most one DMA piece or two fragment reads, pinned with sched_barrier.
    {
        constexpr int NM = (L16 ? 4 : 2) * MT * NTW;  // MFMAs per half K-step (16x16x32: four 16-cycle ones per 32x32 block; else two 32-cycle ones)
        constexpr int NF = 2 * (MT + NTW);  // fragment reads per half K-step
        constexpr int FPM = (NF + NM - 1) / NM < 1 ? 1 : (NF + NM - 1) / NM;   // fragment reads hung behind one MFMA of the first half
        auto frag_ptr = [&](const char* st, int kk0, int f) -> const uint4* {
            const int h = f / (MT + NTW), e = f % (MT + NTW);
            const int slot = frag_slot(kk0, h);
            return reinterpret_cast<const uint4*>(e < MT ? st + a_off[e] + h * HROW + ((slot ^ swz) << 4) : st + w_off[e - MT] + h * HROW + ((slot ^ swzw) << 4));
        };
        // MFMA m of a half K step on fragment set (fa, fw): 32x32x16 form: (h, i, j) - k sub-step h of block (i, j);
        // 16x16x32 form: (i, hr, j, hc) - quad (hr, hc) of block (i, j), row half from fa[hr][i], column half from fw[hc][j]
        auto mfma_m = [&](int m, bf16x8 (&fa)[2][MT], bf16x8 (&fw)[2][NTW]) {
            if constexpr (L16) {
                // plain row-major over the wave's (2 MT) x (2 NTW) grid of 16x16 blocks: srcA (the W fragment) changes with EVERY instruction, srcB (the A-row
                // fragment) stays for 2 NTW of them.  Round 5's serpentine (one operand set changes per instruction: at every row turn srcA stays) and both
                // column-major walks (srcA held for 2 MT instructions) are NOT reproducible on this chip when two workgroups share a CU: with a co-resident
                // workgroup in a bf16 epilogue, 27-30 of 30 launches of a 384-tile GEMM differed (wrong elements: one MFMA's 4th output quarter), order 0: 0 of 30
                // (tools/r06_order_det.sh, profiles/r06_order_det.txt).  The pair loops of the 8-wave tiles never repeat srcA either.
                const int i = m / (4 * NTW), hr = (m / (2 * NTW)) & 1, j = (m >> 1) % NTW, hc = m & 1;
                mfma16_inplace(acc[i][j].q[hr * 2 + hc], fw[hc][j], fa[hr][i]);
            } else {
                const int h = m / (MT * NTW), i = (m / NTW) % MT, j = m % NTW;
                acc[i][j].v = mfma32(fw[h][j], fa[h][i], acc[i][j].v);
            }
        };
        auto frag_store = [&](bf16x8 (&fa)[2][MT], bf16x8 (&fw)[2][NTW], int f, uint4 v) {
            const int h = f / (MT + NTW), e = f % (MT + NTW);
            if (e < MT) fa[h][e] = as_bf16x8(v); else fw[h][e - MT] = as_bf16x8(v);
        };
        unsigned long long c0 = 0, w0 = 0;
        const bool probe = ep.clk_probe && blockIdx.x == 0 && tid == 0;
        if (probe) { c0 = clock64(); w0 = wall_clock64(); }
        auto kstep = [&](int kt, auto more_c, auto dma_c) {
            constexpr bool more = decltype(more_c)::value, dma = decltype(dma_c)::value;
            const char* st = smem + (kt % NS) * STAGE;
            // first half: MFMA(P) with the Q fragment reads spread behind the first MFMAs
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                mfma_m(m, pa, pw);
#pragma unroll
                for (int f = FPM * m; f < FPM * m + FPM; ++f)
                    if (f < NF) frag_store(qa, qw, f, *frag_ptr(st, 2, f));
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (more) {
                // tile kt+1 must have landed; in the steady state tiles kt+2 .. kt+NS-1 stay in flight (loads retire in order)
                if (dma) wait_vmcnt<(NS - 2) * (AJ + WJ)>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kt % NS) * STAGE);
            const char* stn = smem + ((kt + 1) % NS) * STAGE;
            const int kt2 = kmap(kt + NS);   // (HOTK: two K slices serve every step)
            const bf16_t* a_k2 = A + kt2 * BK;  // uniform: tile kt+NS goes into the stage this step just finished reading
            const bf16_t* w_k2 = W + kt2 * BK;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                mfma_m(m, qa, qw);
                constexpr int DP = (AJ + WJ + NM - 1) / NM < 1 ? 1 : (AJ + WJ + NM - 1) / NM;   // DMA pieces per MFMA slot (1 for every product tile)
                constexpr int DSL = (AJ + WJ + DP - 1) / DP;                                      // MFMA slots that carry DMA
                constexpr bool SPREAD = (AJ + WJ <= NM) && (NF <= NM - (AJ + WJ));   // one piece or one read per slot (every product tile)
                if constexpr (SPREAD) {
                    const int pc = dma_piece_of_slot<NM, AJ + WJ>(m);
                    if (dma && pc >= 0) {
                        if (pc < AJ) glds16_sv(a_voff[pc], a_k2, sb + pc * (NW * 1024));
                        else glds16_sv(w_voff[pc - AJ], w_k2, sb + A_BYTES + (pc - AJ) * (NW * 1024));
                    }
                    if (more && pc < 0) {
                        const int f = dma_free_rank<NM, AJ + WJ>(m);
                        if (f < NF) frag_store(pa, pw, f, *frag_ptr(stn, 0, f));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                if (dma) {
#pragma unroll
                    for (int q = 0; q < DP; ++q) {
                        const int pc = m * DP + q;
                        if (pc < AJ) glds16_sv(a_voff[pc], a_k2, sb + pc * (NW * 1024));
                        else if (pc < AJ + WJ) glds16_sv(w_voff[pc - AJ], w_k2, sb + A_BYTES + (pc - AJ) * (NW * 1024));
                    }
                }
                if (more) {
                    constexpr int F0 = (DSL < NM) ? DSL : 0;  // first MFMA slot that carries fragment reads
                    constexpr int PER = (NF + (NM - F0) - 1) / (NM - F0);
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int f = (m - F0) * PER + q;
                        if (m >= F0 && f < NF) frag_store(pa, pw, f, *frag_ptr(stn, 0, f));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ---- kstep_pair (PAIRK: the 8-wave bf16 tiles with two column blocks per wave).  A 16x16 block's two MFMAs of a K step - K half P (k 0..31) and K half
        // Q (k 32..63) - are issued back to back on its accumulator, P first (the summation order per accumulator is the half-by-half loop's: same bits).
        // What makes that possible in the SAME 20 fragment registers, with the same single barrier and the same DMA timing per step:
        //   phase A = the blocks of column block 0 (rows in order, (hc 0: P, Q), (hc 1: P, Q)): 8 MT MFMAs on pa / qa (A rows, both K halves) and pw / qw[.][0];
        //             behind its first MFMAs the four fragments of column block 1 of THIS step are read (their registers died with the previous step's phase B);
        //   barrier   (tile kt + 1 landed; every wave has read everything it needs of stage kt: A rows and column block 0 during the previous step's phase B,
        //             column block 1 just now - so the stage is free for tile kt + NS, exactly as in the half-by-half loop);
        //   phase B = the blocks of column block 1: row r's A-row fragments die with its MFMAs (P after slot 4 r + 2, Q after 4 r + 3) and are re-read for step
        //             kt + 1 right there; the other two slots of a row carry the DMA pieces of tile kt + NS (spread over the phase) and the four fragments of
        //             column block 0 of step kt + 1.
        auto frag_at = [&](const char* stg, int kk0, int h, int e) -> const uint4* {   // fragment half h of A block e (e < MT) or W block e - MT, K half kk0 / 2
            const int slot = frag_slot(kk0, h);
            return reinterpret_cast<const uint4*>(e < MT ? stg + a_off[e] + h * HROW + ((slot ^ swz) << 4) : stg + w_off[e - MT] + h * HROW + ((slot ^ swzw) << 4));
        };
        auto kstep_pair = [&](int kt, auto more_c, auto dma_c) {
            constexpr bool more = decltype(more_c)::value, dma = decltype(dma_c)::value;
            if constexpr (PAIRK) {
            constexpr int NPR = 4 * MT;         // pairs per phase: (2 MT row halves) x (2 column halves of the phase's column block)
            constexpr int ND = AJ + WJ;         // DMA pieces of this wave per tile
            constexpr int NE = 2 * MT;          // phase-B pair slots in mid-row (the row-end ones carry the A-row re-reads)
            const char* st = smem + (kt % NS) * STAGE;
            // One asm statement per pair: hipcc would otherwise give the first MFMA a scratch destination (D != C) whenever that suits its allocation,
            // and only D == C on both instructions is the form the matrix pipe chains without the register file.  The compiler does not see an MFMA here:
            // the wait states a VALU read of the accumulators needs are inserted by hand behind the K loop.
            auto pair_mfma = [&](int pi, int j) {
                const int r = pi >> 1, i = r >> 1, hr = r & 1, hc = pi & 1;
                mfma16_pair(acc[i][j].q[hr * 2 + hc], pw[hc][j], pa[hr][i], qw[hc][j], qa[hr][i]);
            };
#pragma unroll
            for (int pi = 0; pi < NPR; ++pi) {
                pair_mfma(pi, 0);
                if (pi < 2) pw[pi][1] = as_bf16x8(*frag_at(st, 0, pi, MT + 1));
                else if (pi < 4) qw[pi - 2][1] = as_bf16x8(*frag_at(st, 2, pi - 2, MT + 1));
                if (pi < 6) {   // the rows of the second half of the grid (3 .. 5) are read HERE, in the step that uses them (their registers died with the
                    const int r = 3 + (pi >> 1);   // previous step's last pairs): phase B carries 17 items instead of 23
                    if (pi & 1) qa[r & 1][r >> 1] = as_bf16x8(*frag_at(st, 2, r & 1, r >> 1));
                    else pa[r & 1][r >> 1] = as_bf16x8(*frag_at(st, 0, r & 1, r >> 1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (more) {
                if (dma) wait_vmcnt<(NS - 2) * (AJ + WJ)>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kt % NS) * STAGE);
            const char* stn = smem + ((kt + 1) % NS) * STAGE;
            const int kt2 = kmap(kt + NS);
            const bf16_t* a_k2 = A + kt2 * BK;
            const bf16_t* w_k2 = W + kt2 * BK;
            auto piece = [&](int pc) {
                if (dma && pc < ND) {
                    if (pc < AJ) glds16_sv(a_voff[pc], a_k2, sb + pc * (NW * 1024));
                    else glds16_sv(w_voff[pc - AJ], w_k2, sb + A_BYTES + (pc - AJ) * (NW * 1024));
                }
            };
            auto wread = [&](int f) {   // fragment f of column block 0 of step kt + 1
                if (more && f >= 0 && f < 4) {
                    if (f < 2) pw[f][0] = as_bf16x8(*frag_at(stn, 0, f, MT));
                    else qw[f - 2][0] = as_bf16x8(*frag_at(stn, 2, f - 2, MT));
                }
            };
            // Phase-B schedule (MT = 3, 7 pieces): at most ONE DMA piece per pair slot (a piece takes 60-185 cycles to issue: two in a row hold the wave
            // longer than its partner on the SIMD can cover), the re-read of row r's fragments anywhere from its last pair (slot 2 r + 1) on
            constexpr int PC[12] = {0, 1, -1, 2, -1, 3, 4, -1, 5, -1, 6, -1};          // piece of the slot
            constexpr int WR[12] = {0, -1, 1, -1, 2, -1, -1, 3, -1, -1, -1, -1};       // column-block-0 fragment of the slot
            constexpr int AP[12] = {-1, 0, -1, 1, -1, 2, -1, -1, -1, -1, -1, -1};     // row (0 .. 2) whose P-half A fragment is re-read in the slot
            constexpr int AQ[12] = {-1, -1, 0, -1, 1, -1, 2, -1, -1, -1, -1, -1};     // ... Q half
            static_assert(NPR == 12 && ND == 7, "kstep_pair: the phase-B schedule is written for the 192x256 tile");
#pragma unroll
            for (int pi = 0; pi < NPR; ++pi) {
                pair_mfma(pi, 1);
                if (PC[pi] >= 0) piece(PC[pi]);
                if (WR[pi] >= 0) wread(WR[pi]);
                if (more && AP[pi] >= 0) pa[AP[pi] & 1][AP[pi] >> 1] = as_bf16x8(*frag_at(stn, 0, AP[pi] & 1, AP[pi] >> 1));
                if (more && AQ[pi] >= 0) qa[AQ[pi] & 1][AQ[pi] >> 1] = as_bf16x8(*frag_at(stn, 2, AQ[pi] & 1, AQ[pi] >> 1));
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        };
        // kstep_pair1 (PAIRK1: the 8-wave 192x128 tile, one 32-column block per wave): the same turn of the double buffer with the two 16-column HALVES of
        // the block as the phases - phase A = the six pairs of column half 0, phase B = those of half 1.  Six pairs per phase are too few to read rows 3-5
        // in the phase that uses them first: rows 4 and 5 are (four and two pairs ahead), row 3 is re-read in phase B with rows 0-2.
        auto kstep_pair1 = [&](int kt, auto more_c, auto dma_c) {
            constexpr bool more = decltype(more_c)::value, dma = decltype(dma_c)::value;
            if constexpr (PAIRK1) {
            constexpr int ND = AJ + WJ;
            const char* st = smem + (kt % NS) * STAGE;
            auto pair1 = [&](int r, int hc) {
                const int i = r >> 1, hr = r & 1;
                mfma16_pair(acc[i][0].q[hr * 2 + hc], pw[hc][0], pa[hr][i], qw[hc][0], qa[hr][i]);
            };
            auto aread = [&](const char* stg, int r, int q) {   // K half q of row r's A fragment
                if (q) qa[r & 1][r >> 1] = as_bf16x8(*frag_at(stg, 2, r & 1, r >> 1));
                else pa[r & 1][r >> 1] = as_bf16x8(*frag_at(stg, 0, r & 1, r >> 1));
            };
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                pair1(r, 0);
                if (r == 0) { pw[1][0] = as_bf16x8(*frag_at(st, 0, 1, MT)); aread(st, 4, 0); }
                if (r == 1) { qw[1][0] = as_bf16x8(*frag_at(st, 2, 1, MT)); aread(st, 4, 1); }
                if (r == 2) aread(st, 5, 0);
                if (r == 3) aread(st, 5, 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (more) {
                if (dma) wait_vmcnt<(NS - 2) * (AJ + WJ)>(); else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kt % NS) * STAGE);
            const char* stn = smem + ((kt + 1) % NS) * STAGE;
            const int kt2 = kmap(kt + NS);
            const bf16_t* a_k2 = A + kt2 * BK;
            const bf16_t* w_k2 = W + kt2 * BK;
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                pair1(r, 1);
                if (dma && r < ND) {
                    if (r < AJ) glds16_sv(a_voff[r], a_k2, sb + r * (NW * 1024));
                    else glds16_sv(w_voff[r - AJ], w_k2, sb + A_BYTES + (r - AJ) * (NW * 1024));
                }
                if (more) {
                    if (r < 4) { aread(stn, r, 0); aread(stn, r, 1); }
                    if (r == 4) pw[0][0] = as_bf16x8(*frag_at(stn, 0, 0, MT));
                    if (r == 5) qw[0][0] = as_bf16x8(*frag_at(stn, 2, 0, MT));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        };
        auto kstep_mx = [&](int kt, auto more_c, auto dma_c) {
            constexpr bool more = decltype(more_c)::value, dma = decltype(dma_c)::value;
            constexpr int NX = MT * NTW;  // scaled MFMAs per half K-step (64 cycles each: the same pipe time as the 2 NX bf16 ones)
            const char* st = smem + (kt % NS) * STAGE;
            // first half: k 0..63 of the step = the P fragments (both slots of each operand), k-blocks 0 | 1 -> op_sel 0
#pragma unroll
            for (int m = 0; m < NX; ++m) {
                const int i = m / NTW, j = m % NTW;   // (plain row-major: consecutive MFMAs never share srcA, see mfma_m)
                if constexpr (!L16) acc[i][j].v = mfma_mx<0>(pw[0][j], pw[1][j], pa[0][i], pa[1][i], acc[i][j].v, scw[j], sca[i]);
#pragma unroll
                for (int f = 2 * m; f < 2 * m + 2; ++f)
                    if (f < NF) frag_store(qa, qw, f, *frag_ptr(st, 2, f));
                __builtin_amdgcn_sched_barrier(0);
            }
            static_assert(!FP8 || NF <= 2 * NX, "two fragment reads per scaled MFMA cover the Q set");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (more) {
                wait_vmcnt<0>();  // NS == 2: tile kt+1 (and its scale piece) landed
                __builtin_amdgcn_s_barrier();
            }
            const unsigned sb = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(kt % NS) * STAGE);
            const char* stn = smem + ((kt + 1) % NS) * STAGE;
            const bf16_t* a_k2 = A + (kt + NS) * BK;
            const bf16_t* w_k2 = W + (kt + NS) * BK;
            unsigned nsa[MT], nsw[NTW];
            // second half: k 64..127 = the Q fragments, k-blocks 2 | 3 -> op_sel 2; two DMA pieces per MFMA slot first, then the next
            // step's P fragments and scale words
            constexpr int ND = AJ + WJ;                  // tile pieces of this wave (+ one scale piece on waves 0 / 1)
            constexpr int DS = (ND + 1 + 1) / 2;         // MFMA slots that carry DMA
            static_assert(!FP8 || DS < NX, "DMA pieces must leave MFMA slots for the fragment reads");
            constexpr int RS = NX - DS > 0 ? NX - DS : 1;  // (only meaningful for the FP8 instantiations)
            constexpr int PER = (NF + RS - 1) / RS;
#pragma unroll
            for (int m = 0; m < NX; ++m) {
                const int i = m / NTW, j = m % NTW;
                if constexpr (!L16) acc[i][j].v = mfma_mx<2>(qw[0][j], qw[1][j], qa[0][i], qa[1][i], acc[i][j].v, scw[j], sca[i]);
                if (dma) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int pc = 2 * m + q;
                        if (pc < AJ) glds16_sv(a_voff[pc], a_k2, sb + pc * (NW * 1024));
                        else if (pc < ND) glds16_sv(w_voff[pc - AJ], w_k2, sb + A_BYTES + (pc - AJ) * (NW * 1024));
                        else if (pc == ND) issue_scales(kt + NS);
                    }
                }
                if (more && m >= DS) {
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        const int f = (m - DS) * PER + q;
                        if (f < NF) frag_store(pa, pw, f, *frag_ptr(stn, 0, f));
                    }
                    if (m == NX - 1) {
                        const char* sc = stn + A_BYTES + W_BYTES;
#pragma unroll
                        for (int i2 = 0; i2 < MT; ++i2) nsa[i2] = *reinterpret_cast<const unsigned*>(sc + (wm * (MT * 32) + i2 * 32 + frow) * 4) >> (8 * fhalf);
#pragma unroll
                        for (int j2 = 0; j2 < NTW; ++j2) nsw[j2] = *reinterpret_cast<const unsigned*>(sc + 1024 + (wn * (NTW * 32) + j2 * 32 + frow) * 4) >> (8 * fhalf);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) {
#pragma unroll
                for (int i2 = 0; i2 < MT; ++i2) sca[i2] = nsa[i2];
#pragma unroll
                for (int j2 = 0; j2 < NTW; ++j2) scw[j2] = nsw[j2];
            }
        };
        {
            using T = std::integral_constant<bool, true>;
            using F = std::integral_constant<bool, false>;
            int kt = 0;
            if constexpr (FP8) {
                for (; kt + NS < nk; ++kt) kstep_mx(kt, T{}, T{});
                for (; kt + 1 < nk; ++kt) kstep_mx(kt, T{}, F{});
                kstep_mx(kt, F{}, F{});
            } else {
            if constexpr (PAIRK1) {
                for (; kt + NS < nk; ++kt) kstep_pair1(kt, T{}, T{});
                for (; kt + 1 < nk; ++kt) kstep_pair1(kt, T{}, F{});
                kstep_pair1(kt, F{}, F{});
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");   // (see below)
            } else if constexpr (PAIRK) {
                for (; kt + NS < nk; ++kt) kstep_pair(kt, T{}, T{});
                for (; kt + 1 < nk; ++kt) kstep_pair(kt, T{}, F{});
                kstep_pair(kt, F{}, F{});
                // (the accumulators were last written by asm-issued MFMAs hipcc does not know as such: the wait states a VALU read of a 8-pass XDL result needs)
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
            } else {
            for (; kt + NS < nk; ++kt) kstep(kt, T{}, T{});     // steady state: branch-free
            for (; kt + 1 < nk; ++kt) kstep(kt, T{}, F{});       // the last NS-1 K steps but one: nothing left to prefetch
            kstep(kt, F{}, F{});                                 // last K step
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");        // (asm-issued MFMAs here too: mfma16_inplace)
            }
            }
        }
        if (probe && (int)blockIdx.y == ep.kparts - 1) {
            g_clk_probe[0] = clock64() - c0;
            g_clk_probe[1] = wall_clock64() - w0;
            g_clk_probe[2] = (unsigned long long)nk;
        }
        unsigned long long e0 = 0;
        if (probe) e0 = clock64();
        if constexpr (MODE == 2 && !PERS && !FP8) {
            if (ep.sk_ord) {
                // ordered split-K: wait until the parts before this one have added their share of this tile to H.  A part's
                // workgroups are dispatched after every workgroup of the parts before it (blockIdx.y is the slow grid index), so
                // whoever is waited for is already running.  The counter lives in L2 like the fp32 atomics it replaces.
                if (tid == 0) {
                    const int tile = tm * tiles_n + tn;
                    // (bounded: a counter left behind by a faulted launch must not hang the device; after ~1 s the update goes ahead and the
                    //  missed turn is COUNTED: the host turns it into an error and re-zeroes the counters, gemm_splitk_poll)
                    int spins = 0;
                    for (; atomicAdd(ep.sk_cnt + tile, 0) != (int)blockIdx.y && spins < (1 << 20); ++spins) __builtin_amdgcn_s_sleep(8);
                    if (spins >= (1 << 20)) atomicAdd(ep.sk_cnt + SK_MAX_TILES, 1);
                }
                asm volatile("" ::: "memory");
            }
        }
        gemm_epilogue<MODE, MT, NTW, !FP8, L16>(acc, smem, Cv, ldc, M, N, ep, m0, n0, wm, wn, vw, lane, BMv, BNv, (VEC && vec_on) ? smem + VOFF : nullptr);
        if constexpr (MODE == 2 && !PERS && !FP8) {
            if (ep.sk_ord) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this part's H rows are in L2
                __syncthreads();
                if (tid == 0) atomicExch(ep.sk_cnt + tm * tiles_n + tn, (int)blockIdx.y + 1 == ep.ksplit ? 0 : (int)blockIdx.y + 1);
            }
        }
        if (probe) {
            g_clk_probe[3] = clock64() - e0;            // epilogue until the last store is ISSUED
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            g_clk_probe[4] = clock64() - e0;            // ... until the stores are acknowledged
            g_clk_probe[5] = c0 - t_entry;              // tile map + prologue (first two K tiles landed, first fragments read)
        }
        if (!PERS) return;
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();  // every wave has read back its staging slice: the next tile's DMA may overwrite it
        continue;
    }

    }
}


// Split-K placement probe: every workgroup of a (16, 2) grid reports the XCD it runs on (HW_REG_XCC_ID, bits 3:0).
__global__ void xcd_probe_kernel(int* out) {
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf);
}

}  // namespace

// The split-K parts of a tile (same blockIdx.x, consecutive blockIdx.y) exchange their share of H through L2, which is private to an
// XCD: that is only right while both parts run on the same XCD.  The dispatcher places workgroup b on XCD b % 8 and the grids here
// have gridDim.x % 8 == 0, but HIP promises no placement - so it is CHECKED once per process on the device in use (a (16, 2) probe grid:
// same XCD for both blockIdx.y of every blockIdx.x, and XCD == blockIdx.x % 8).  Not verified (not run, or the check failed): launch_gemm
// does not split K at all.  Called from the handle constructors (never under stream capture).
static int g_splitk_ok = -1;  // -1 unknown, 0 refused, 1 verified
int gemm_verify_splitk_placement() {
    if (g_splitk_ok >= 0) return 0;
    int* d = nullptr;
    int h[32];
    ACE_HIP(hipMalloc((void**)&d, sizeof(h)));
    hipLaunchKernelGGL(xcd_probe_kernel, dim3(16, 2), dim3(64), 0, nullptr, d);
    hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return hip_fail(e, "split-K placement probe", __FILE__, __LINE__);
    bool ok = true;
    for (int x = 0; x < 16; ++x) ok = ok && h[x] == h[16 + x] && h[x] == (h[0] + x) % 8;
    g_splitk_ok = ok ? 1 : 0;
    if (!ok) fprintf(stderr, "[ace355] workgroup -> XCD placement is not blockIdx %% 8 on this device: split-K launches disabled\n");
    return 0;
}

int gemm_splitk_poll(int* sk_cnt, hipStream_t s) {
    if (!sk_cnt) return 0;
    int missed = 0;
    ACE_HIP(hipMemcpyAsync(&missed, sk_cnt + SK_MAX_TILES, sizeof(int), hipMemcpyDeviceToHost, s));
    ACE_HIP(hipStreamSynchronize(s));
    if (missed) {
        ACE_HIP(hipMemsetAsync(sk_cnt, 0, SK_CNT_INTS * sizeof(int), s));
        set_error("gemm: " + std::to_string(missed) + " ordered split-K turn(s) timed out; the result of this call is invalid (counters reset; "
                  "ACE355_GEMM_SKORD=0 selects the atomic path)");
        return 2;
    }
    return 0;
}

static int gemm_variant() {
    static int v = -1;
    if (v < 0) {
        // ACE355_GEMM=v1 selects the simple register-staged kernel (A/B + bring-up reference); default = LDS-DMA pipeline
        const char* e = getenv("ACE355_GEMM");
        v = (e && e[0] == 'v' && e[1] == '1') ? 1 : 4;
    }
    return v;
}
bool gemm_fold_supported() { return gemm_variant() != 1; }
static bool variant_is_v1() { return gemm_variant() == 1; }
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// The CUs a GEMM launch plans for: GemmEpilogue::cu_slots, else ACE355_MAX_WGS (default 256 = the chip).  With 128 a launch is shaped for
// HALF the chip (tile choice, persistent grid of 128 workgroups), so that two independent launch sequences on two hardware queues run
// side by side on 128 CUs each instead of interleaving their workgroups over all 256 (the dual-chain sampler, dit.hip).
static int gemm_cu_slots(int hint = 0) {
    static int v = -1;
    if (v < 0) {
        v = env_int("ACE355_MAX_WGS", 256);
        if (v < 8 || v > 256) v = 256;
        v &= ~7;
    }
    if (hint >= 8 && hint <= 256) return hint & ~7;   // GemmEpilogue::cu_slots
    return v;
}

static int choose_xcd_m(int tiles_m, int tiles_n, int M, int N, int xcd_m_env) {
    int xcd_m = 8;
    double best = 1e30;
    for (int xm = 1; xm <= 8; xm *= 2) {
        const int xn = 8 / xm;
        if (xm > tiles_m || xn > tiles_n) continue;
        const double cost = (double)xn * M + (double)xm * N;
        if (cost < best) { best = cost; xcd_m = xm; }
    }
    if (xcd_m_env == 1 || xcd_m_env == 2 || xcd_m_env == 4 || xcd_m_env == 8) xcd_m = xcd_m_env;
    return xcd_m;
}

template <int MODE>
static void launch_mode(int variant, int mt, int big, hipStream_t s, const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc,
                        int M, int N, int K, const GemmEpilogue& ep, int tiles_n, int nwg) {
    if constexpr (MODE != 4) {
        if (variant == 1) {
            hipLaunchKernelGGL(gemm_kernel<MODE>, dim3(nwg), dim3(256), 0, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg);
            return;
        }
    }
    static int group_m = -1, xcd_m_env = -1, pers = 0, deep_env = 0;
    if (group_m < 0) {
        deep_env = env_int("ACE355_GEMM_DEEP", -1);    // -1 heuristic, 0 never, 1 always (4-wave tiles)
        pers = env_int("ACE355_GEMM_PERS", 1);         // persistent workgroups for multi-round launches
        group_m = env_int("ACE355_GEMM_GROUPM", 4);    // rasterisation group height
        xcd_m_env = env_int("ACE355_GEMM_XCDM", 0);    // pin the XCD grid shape
    }
    // XCD grid: minimise xcd_n*|A| + xcd_m*|W| = (8/xm) * M + xm * N (same K), over xm in {1,2,4,8}
    const int tiles_m = nwg / tiles_n;
    const int xcd_m = choose_xcd_m(tiles_m, tiles_n, M, N, xcd_m_env);
    const int xcd_n = 8 / xcd_m;
    const int region = ((tiles_m + xcd_m - 1) / xcd_m) * ((tiles_n + xcd_n - 1) / xcd_n);
    // (+ GemmEpilogue::pf_x prefetch workgroups per XCD behind the region's slots: launch_gemm cleared pf_x where they do not belong)
    const dim3 grid(8 * (region + (ep.pf_x > 0 ? ep.pf_x : 0)), ep.kparts > 1 ? ep.kparts : 1);
    // one workgroup per CU at most: nothing but a deeper DMA pipeline hides the weight stream's HBM latency
    const int cus = gemm_cu_slots(ep.cu_slots), pers_x = cus / 8;   // persistent workgroups per XCD
    const bool deep = deep_env >= 0 ? deep_env != 0 : (long)nwg * (ep.kparts > 1 ? ep.kparts : 1) <= cus;
#define ACE_LAUNCH_SP(kern, thr) hipLaunchKernelGGL(kern, grid, dim3(thr), 0, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg, group_m, xcd_m)
    if constexpr (MODE == 4) {  // head epilogue: every tile whose N-waves pair up over a 128-column head (not the 2-stage mid tile, not v1)
        if (big == 2) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 4, 1, 0, 3>), 512);
        else if (big && pers && region > pers_x) {
            const dim3 pgrid(8 * pers_x);
            hipLaunchKernelGGL((gemm_sp_kernel<MODE, 3, 4, 2, 1>), pgrid, dim3(512), 0, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg, group_m,
                               xcd_m);
        } else if (big) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 4>), 512);
        else if (mt == 1) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 1, 2, 2, 0, 4>), 256);
        else if (deep && mt == 3) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 2, 2, 0, 3>), 256);
        else if (deep) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 2, 2, 2, 0, 4>), 256);
        else if (mt == 3) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 2>), 256);
        else ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 2, 2>), 256);
    } else if (big == 2) {
        static int mid_ns = -1;
        if (mid_ns < 0) mid_ns = env_int("ACE355_GEMM_MIDNS", 3);
        if constexpr (MODE != 3) {  // 192x128, 8 waves (wave tile 96x32); one workgroup per CU: 3 stages of 40 KB
            if (mid_ns == 3) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 4, 1, 0, 3>), 512);
            else ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 4, 1>), 512);
        }
    } else if (big && pers && region > pers_x && (MODE != 2 || pers >= 2)) {
        const dim3 pgrid(8 * pers_x);
        hipLaunchKernelGGL((gemm_sp_kernel<MODE, 3, 4, 2, 1>), pgrid, dim3(512), 0, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg, group_m,
                           xcd_m);
    } else if (big) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 4>), 512);
    else if (mt == 1) {
        if constexpr (MODE != 3) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 1, 2, 2, 0, 4>), 256);  // 64 x 128, 4 x 24 KB stages
    }
    else if (deep && mt == 3) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 2, 2, 0, 3>), 256);   // 3 x 40 KB stages
    else if (deep) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 2, 2, 2, 0, 4>), 256);              // 4 x 32 KB stages
    else if (mt == 3) ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 3, 2>), 256);
    else ACE_LAUNCH_SP((gemm_sp_kernel<MODE, 2, 2>), 256);
#undef ACE_LAUNCH_SP
}

static int g_k_rotation = -1;   // ace355_gemm_set_k_rotation / ACE355_GEMM_KROT (default 1)
static int k_rotation_mode() {
    if (g_k_rotation < 0) g_k_rotation = env_int("ACE355_GEMM_KROT", 1);
    return g_k_rotation;
}
int gemm_k_rotation_mode() { return k_rotation_mode(); }
int gemm_set_k_rotation(int mode) {
    const int prev = k_rotation_mode();
    g_k_rotation = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    return prev;
}

struct TileChoice { int mt, bn, big; };
static TileChoice choose_tile(int M, int N, int mode, int cu_hint, int variant) {
    // Tile choice.  The kernel is L2->LDS bandwidth bound (ablation in DESIGN.md), so the biggest tile that still fills
    // the chip wins: 192x256 (8 waves, 110 flop per staged byte) when it yields >= ~0.8 x 256 workgroups, else the
    // 4-wave 128/192 x 128 tiles with the height that minimises (rounds of 512 resident workgroups) x rows.
    int mt = 2, bn = BN;
    int big = 0;  // 0: 4-wave 128/192x128 tiles, 1: 8-wave 192x256, 2: 8-wave 192x128
    if (variant != 1) {
        const long cus = gemm_cu_slots(cu_hint);
        const long slots = 2 * cus;
        const long tn128 = (N + 127) / 128;
        const long t128 = (long)((M + 127) / 128) * tn128, t192 = (long)((M + 191) / 192) * tn128;
        const long c128 = ((t128 + slots - 1) / slots) * 128, c192 = ((t192 + slots - 1) / slots) * 192;
        if (c192 < c128) mt = 3;
        static int bigenv = -1;
        if (bigenv < 0) bigenv = env_int("ACE355_GEMM_BIG", 1);  // 0 never, 1 heuristic, 2 always
        const long tbig = (long)((M + 191) / 192) * ((N + 255) / 256);
        // (>= 180 tiles: three quarters of the CUs with one 8-wave workgroup each beat the same work as 384 four-wave workgroups on 512
        //  slots - the SwiGLU projection of a batch-1 request, M = 750: 38.5 vs 42.4 us, round 3)
        if (bigenv == 2 || (bigenv == 1 && tbig >= 180 * cus / 256 && (N % 256 == 0 || N >= 1024))) { mt = 3; bn = 256; big = 1; }
        else if (mode != 3 && (bigenv == 3 || (bigenv == 1 && t192 >= 200 * cus / 256 && t192 <= 320 * cus / 256))) { mt = 3; bn = 128; big = 2; }
    }
    // One sequence's worth of rows (the conditional rows' cross-attention projections of a one-song request: M = 375 -> 48 workgroups of
    // 128 x 128, each pulling 32 KB per K step through ONE CU's DMA path while 200 CUs idle): 64-row tiles double the workgroups and cut a
    // K step to 24 KB (ACE355_GEMM_MT1=0 for A/B).  Only where the 128-row form leaves more than half of the chip without a workgroup.
    if (variant != 1 && big == 0 && mode != 3) {
        static int mt1_env = -1;
        if (mt1_env < 0) mt1_env = env_int("ACE355_GEMM_MT1", 1);
        const long cus = gemm_cu_slots(cu_hint);
        const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
        if (mt1_env && t128 * 2 <= cus && M > 64) mt = 1;
    }
    return TileChoice{mt, bn, big};
}

// GemmEpilogue::pf_* for a launch that PRECEDES launch_gemm(.., W, M, N, K, mode ..) in the caller's sequence (declared in common.h).
// Where it pays (round 6, tools/r06_gemm_pf_ab*.sh, tools/r06_hot_cold_w*.py; one 30 s song, DiT only, same box): 135.2-135.7 -> 129.8-131.0 ms with
// 16-32 prefetch workgroups per XCD and 1-4 MB read per XCD region (all within noise of each other; 2-8 workgroups make the prefetch the
// launch's critical path: 145-208 ms); per launch the head-norm projections gain what a back-to-back repeat of the launch gains (QKV 26.8 ->
// 23.1 us, cross-q 18.8 -> 15.1), the residual projections ~ 1 us, gate|up nothing (6.3 MB per XCD do not fit beside the running launch's
// working set: left out).  Requests of two or more songs lose (187 -> 191 ms at one chain of two songs): one sequence's rows only.
void gemm_prefetch_plan(GemmEpilogue* ep, const void* W, int M, int N, int K, int mode, int cu_slots) {
    constexpr int PF_X = 24;            // prefetch workgroups per XCD
    constexpr int PF_CAP_KB = 2048;     // bytes of a region that are read, per XCD (an L2 holds 4 MB)
    constexpr int PF_MAX_ROWS = 800;    // token rows of the next launch (one 30 s song with its CFG copy: 750)
    static int on = -1, xcd_m_env = 0;
    if (on < 0) {
        on = env_int("ACE355_GEMM_PF", 1);   // 0: no prefetch workgroups anywhere (A/B, tests)
        xcd_m_env = env_int("ACE355_GEMM_XCDM", 0);
    }
    const int variant = gemm_variant();
    if (!on || variant == 1 || !W || M > PF_MAX_ROWS || K % 8 || !(mode == 2 || mode == 4)) return;
    const TileChoice tc = choose_tile(M, N, mode, cu_slots, variant);
    const int tiles_n = (N + tc.bn - 1) / tc.bn, tiles_m = (M + tc.mt * 64 - 1) / (tc.mt * 64);
    const int xcd_m = choose_xcd_m(tiles_m, tiles_n, M, N, xcd_m_env), xcd_n = 8 / xcd_m;
    const int rn = (tiles_n + xcd_n - 1) / xcd_n;
    const unsigned long long chunk16 = (unsigned long long)rn * tc.bn * K / 8, total16 = (unsigned long long)N * K / 8;
    if (total16 >= (1ull << 32)) return;
    ep->pf_w = W;
    ep->pf_chunk16 = (unsigned)std::min(chunk16, total16);
    ep->pf_total16 = (unsigned)total16;
    ep->pf_len16 = (unsigned)std::min<unsigned long long>(ep->pf_chunk16, (unsigned long long)PF_CAP_KB * 64ull);
    ep->pf_xcd_n = xcd_n;
    ep->pf_x = PF_X;
}

int launch_gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M, int N, int K,
                const GemmEpilogue& ep_in, hipStream_t s) {
    GemmEpilogue ep = ep_in;
    {
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const int per16 = (ep.mode == 0 || ep.mode == 3 || ep.mode == 4) ? 8 : 4;  // elements of C per 16 bytes
        ep.wide_ok = al16(C) && (ldc % per16) == 0 && al16(ep.bias) && al16(ep.g1) && al16(ep.g2) && al16(ep.cvec) &&
                     (ep.g2_stride % 4) == 0;
        if (env_int("ACE355_GEMM_SCALAR_EPI", 0)) ep.wide_ok = 0;  // A/B + test hook
        static int clk = -1;
        if (clk < 0) clk = env_int("ACE355_GEMM_CLK", 0);
        ep.clk_probe = clk;
        const int krot = k_rotation_mode();   // 1: launches with N <= 2048, 2: every launch (the kernel applies it to one-round bf16 launches)
        ep.krot = (krot == 2 || (krot == 1 && N <= 2048)) && (K / 64) % 8 == 0;
        static int rowst = -1;
        if (rowst < 0) rowst = std::min(std::max(env_int("ACE355_GEMM_KROT_ROW", 1), 0), 15);   // row stagger in K steps (with the rotation only); 1 since the pair K loop (0: - 0.3 %, 2: + 0.5 %)
        if (ep.krot) ep.krot |= rowst << 4;
    }
    ACE_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem");
    ACE_CHECK(K % BK == 0, "gemm: K must be a multiple of 64");
    ACE_CHECK((long)std::max(M, ep.a_zero_idx + 1) * lda < (1L << 31) && (long)N * ldw < (1L << 31), "gemm: A and W must each be smaller than 4 GB (32-bit DMA offsets)");
    ACE_CHECK(ep.a_zero_idx == 0 || ep.a_zero_idx >= M, "gemm: the zero row of A lies behind its M rows");
    if (variant_is_v1()) ep.a_zero_idx = 0;
    ACE_CHECK(ep.a_wrap == 0 || (!variant_is_v1() && ep.a_wrap > 0 && 2 * ep.a_wrap >= M), "gemm: a_wrap needs the DMA kernels and at most two copies of the rows");
    ACE_CHECK((lda % 8) == 0 && (ldw % 8) == 0, "gemm: lda/ldw must be multiples of 8 (16-B rows)");
    ACE_CHECK(ep.mode != 3 || (N % 64) == 0, "gemm: swiglu needs N % 64 == 0");
    ACE_CHECK(ep.mode != 2 || !ep.g1 || ep.rows_per_seq > 0, "gemm: rows_per_seq must be > 0");
    const int variant = gemm_variant();
    const TileChoice tc = choose_tile(M, N, ep.mode, ep.cu_slots, variant);
    const int mt = tc.mt, bn = tc.bn, big = tc.big;
    // (Slab split-K - K cut over blockIdx.y for EVERY mode, partial accumulators parked in an XCD's L2 and reduced by the last part - was built in
    //  round 3, measured slower than the deep pipeline it competes with (157.4 vs 149.9 ms per one-song request) and removed in round 6:
    //  tools/r06_slab_splitk.patch)
    ep.kparts = 1;
    const int tiles_n = (N + bn - 1) / bn;
    const int bm = mt * 64;
    const int tiles_m = (M + bm - 1) / bm;
    const int nwg = tiles_m * tiles_n;
    // Small-M residual GEMMs (batch-1 requests: M = 750 rows -> 96 workgroups, each alone with a 32-96 step K loop) are
    // latency bound: split K over blockIdx.y and let every part add gate * partial into H.  Only mode 2 (its epilogue is an
    // accumulation already).  With turn counters from the caller (ep.sk_cnt) the parts add in part order with plain read-modify-writes
    // - bit-reproducible, and two serialised 4 us updates cost less than 16 k fp32 atomics per part (12 us); without counters the
    // parts use fp32 atomics and the last bits depend on the arrival order.  ACE355_GEMM_KSPLIT=1 disables the split.
    ep.ksplit = 1;
    ep.sk_ord = 0;
    // (K rotation mode 0 = "one summation order whatever the launch shape": no split-K either - a one-song launch then adds up a row's K range
    //  exactly as the same row inside a batch of 8 does)
    if (variant != 1 && ep.mode == 2 && big == 0 && g_splitk_ok == 1 && k_rotation_mode() != 0) {
        static int ks_env = -1;
        if (ks_env < 0) ks_env = env_int("ACE355_GEMM_KSPLIT", 0);
        const int nk = K / BK;
        int ks = 1;
        if (nwg < 128) {  // stay at one workgroup per CU: that regime gets the deep (3-4 stage) pipeline, measured best together
            while (ks < 8 && nwg * ks * 2 <= 256 && nk / (ks * 2) >= 4) ks *= 2;
        }
        static int ord_env = -1;
        if (ord_env < 0) ord_env = env_int("ACE355_GEMM_SKORD", 1);  // 0: fp32 atomics in arrival order (A/B)
        const bool ord = ord_env && ep.sk_cnt && nwg <= SK_MAX_TILES;
        if (ord && ks > 2) ks = 2;   // the parts' read-modify-writes of a tile run one after the other: measured best at two parts
        if (ks_env >= 1) ks = ks_env;
        ks = std::min(ks, nk);
        while (ks > 1 && (ks - 1) * ((nk + ks - 1) / ks) >= nk) --ks;   // every part owns at least one K step (an ordered part must arrive)
        ep.ksplit = ks;
        ep.kparts = ks;
        ep.sk_ord = (ord && ks > 1) ? 1 : 0;
    }
    if ((ep.nf_xg || ep.nc_rowsq) && ep.ksplit > 1 && !ep.sk_ord) {
        // a folded-norm producer needs the FINISHED h in one part's hands: with the ordered turns that is the last part, with fp32 atomics
        // in arrival order (ACE355_GEMM_SKORD=0, or no counters lent) nobody - such a launch keeps its K range whole (advisor r3)
        ep.ksplit = 1;
        ep.kparts = 1;
    }
    if (ep.nf_xg || ep.nc_rowsq) {  // folded RMSNorm (dit.hip): lives in the wide epilogue only, whole tiles, no split-K
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        ACE_CHECK(variant != 1 && ep.wide_ok && N % bn == 0 && (ep.ksplit == 1 || ep.sk_ord), "gemm: the folded-norm epilogue needs whole, 16-byte aligned tiles");
        ACE_CHECK(!ep.nf_xg || (ep.mode == 2 && ep.nf_gA && ep.nf_gB && ep.nf_sqA && ep.nf_sqB && ep.nf_ldx % 8 == 0 && al16(ep.nf_xg) &&
                                al16(ep.nf_gA) && al16(ep.nf_gB)), "gemm: folded-norm producer arguments");
        ACE_CHECK(!ep.nc_rowsq || ((ep.mode == 0 || ep.mode == 3 || ep.mode == 4) && al16(ep.nc_bias)), "gemm: folded-norm consumer arguments");
    }
    // prefetch workgroups for the next launch's weights (GemmEpilogue::pf_*) ride only on launches that leave CUs without a tile
    if (ep.pf_x > 0 && (variant == 1 || !ep.pf_w || (long)nwg * ep.kparts > gemm_cu_slots(ep.cu_slots))) ep.pf_x = 0;
    switch (ep.mode) {
        case 0: launch_mode<0>(variant, mt, big, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg); break;
        case 1: launch_mode<1>(variant, mt, big, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg); break;
        case 2: launch_mode<2>(variant, mt, big, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg); break;
        case 3: launch_mode<3>(variant, mt, big, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg); break;
        case 4: {
            static int fuse = -1;
            if (fuse < 0) fuse = env_int("ACE355_GEMM_HEADEPI", 1);  // 0: always mode 0 + the standalone kernel (A/B runs)
            ACE_CHECK(ep.hn_wq && ep.hn_wk && (!ep.hn_cos == !ep.hn_sin) && ep.rows_per_seq > 0 && ep.hn_q_cols % 128 == 0 &&
                      ep.hn_qk_cols % 128 == 0 && ep.hn_qk_cols <= N && !ep.bias, "gemm: head epilogue arguments");
            static int mid_ns4 = -1;
            if (mid_ns4 < 0) mid_ns4 = env_int("ACE355_GEMM_MIDNS", 3);
            const int tw = big == 1 ? 256 : 128;  // tile width: q | k | v boundaries must fall on tile edges
            const bool fused = fuse && variant != 1 && (big != 2 || mid_ns4 == 3) && ep.wide_ok && N % tw == 0 && ep.hn_q_cols % tw == 0 &&
                               ep.hn_qk_cols % tw == 0;
            static int vt_env = -1;
            if (vt_env < 0) vt_env = env_int("ACE355_GEMM_VT", 2);   // 0: always the separate transpose_v launch (A/B); 1: not on the 192x256 tiles
            // V^T from the epilogue.  Round 3 kept it to the small-M launches; on the big persistent tiles it is free as well: the v tiles'
            // epilogue is a plain store (4 k cycles against 17 k for the q / k tiles' head-norm + RoPE) and the XCDs that own the v columns
            // finish ~9 us before the others, so their 2-byte V^T stores hide there and the transpose_v launch goes (- 4.9 ms per 8-song pass)
            if (!(fused && vt_env && (big != 1 || vt_env >= 2) && ep.vt_out && ep.vt_ld > 0 && ep.vt_heads > 0 && (N - ep.hn_qk_cols) == ep.vt_heads * 128)) ep.vt_out = nullptr;
            if (ep.vt_done) *ep.vt_done = ep.vt_out ? 1 : 0;
            if (fused) {
                launch_mode<4>(variant, mt, big, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg);
            } else {  // two kernels; with a table the q / k columns are in head-pair order (PACK_ROWS_HEADPAIR), without in plain order
                ep.mode = 0;
                launch_mode<0>(variant, mt, big, s, A, lda, W, ldw, C, ldc, M, N, K, ep, tiles_n, nwg);
                ACE_LAUNCH_CHECK();
                const int rc = launch_headnorm_rope2(reinterpret_cast<bf16_t*>(C), M, ldc, 0, ep.hn_qk_cols / 128, ep.hn_wq, ep.hn_wk,
                                                     ep.hn_q_cols / 128, ep.hn_eps, ep.hn_cos, ep.hn_sin, ep.rows_per_seq, s,
                                                     /*paired*/ ep.hn_cos ? 1 : 0);
                if (rc) return rc;
            }
            break;
        }
        default: ACE_CHECK(false, "gemm: bad epilogue mode");
    }
    ACE_LAUNCH_CHECK();
    if (ep.clk_probe) {
        unsigned long long h[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        ACE_HIP(hipStreamSynchronize(s));
        ACE_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk_probe), sizeof(h)));
        if (h[1]) fprintf(stderr, "[ace355 gemm clk] M=%d N=%d K=%d mode=%d kparts=%d tiles=%d: %.3f GHz shader clock, %.0f cycles / K-step (%.3f us) x %llu; prologue %.0f, "
                          "epilogue issue %.0f / acked %.0f cycles (last tile of workgroup 0)%s\n", M, N, K, ep.mode, ep.kparts, nwg,
                          (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / (double)h[2], (double)h[1] * 0.01 / (double)h[2], h[2],
                          (double)h[5], (double)h[3], (double)h[4], "");
        if (h[1] && (ep.mode == 4 || ep.mode == 0 || ep.mode == 3) && h[10]) fprintf(stderr, "[ace355 gemm clk]   epilogue phases (wave 0): sums exchanged %llu, staged %llu cycles\n", h[9], h[10]);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ MXFP8 entry
bool gemm_mx_supported(int M, int N, int K, int mode) {
    return (mode == 0 || mode == 2 || mode == 3 || mode == 4) && M >= 1 && K % 128 == 0 && N % 256 == 0 && K >= 256;
}

template <int MODE>
static void launch_mx_mode(hipStream_t s, const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M, int N, int Kh,
                           const GemmEpilogue& ep, int tiles_n, int nwg) {
    static int group_m = -1, pers = 1;
    if (group_m < 0) {
        pers = env_int("ACE355_GEMM_PERS", 1);
        group_m = env_int("ACE355_GEMM_GROUPM", 4);
    }
    const int tiles_m = nwg / tiles_n;
    int xcd_m = 8;
    double best = 1e30;
    for (int xm = 1; xm <= 8; xm *= 2) {
        const int xn = 8 / xm;
        if (xm > tiles_m || xn > tiles_n) continue;
        const double cost = (double)xn * M + (double)xm * N;
        if (cost < best) { best = cost; xcd_m = xm; }
    }
    const int xcd_n = 8 / xcd_m;
    const int region = ((tiles_m + xcd_m - 1) / xcd_m) * ((tiles_n + xcd_n - 1) / xcd_n);
    // (the persistent residual variant does not fit 256 VGPRs with the scale registers: 60 spilled; its launches are one round anyway)
    const int pers_x = gemm_cu_slots(ep.cu_slots) / 8;
    if (pers && region > pers_x && MODE != 2)
        hipLaunchKernelGGL((gemm_sp_kernel<MODE, 3, 4, 2, 1, 2, 1>), dim3(8 * pers_x), dim3(512), 0, s, A, lda, W, ldw, C, ldc, M, N, Kh, ep, tiles_n, nwg,
                           group_m, xcd_m);
    else
        hipLaunchKernelGGL((gemm_sp_kernel<MODE, 3, 4, 2, 0, 2, 1>), dim3(8 * region), dim3(512), 0, s, A, lda, W, ldw, C, ldc, M, N, Kh, ep, tiles_n,
                           nwg, group_m, xcd_m);
}

int launch_gemm_mx(const uint8_t* Aq, const uint32_t* sa, int sa_ld, const uint8_t* Wq, const uint32_t* sw, int sw_ld, void* C, int ldc,
                   int M, int N, int K, const GemmEpilogue& ep_in, hipStream_t s) {
    ACE_CHECK(Aq && Wq && sa && sw && C, "gemm_mx: null pointer");
    ACE_CHECK(gemm_mx_supported(M, N, K, ep_in.mode), "gemm_mx: unsupported shape / mode (K % 128, N % 256, modes 0 2 3 4)");
    ACE_CHECK(sa_ld >= ((M + 191) / 192) * 192 + 64 && sw_ld >= N && (sa_ld % 4) == 0 && (sw_ld % 4) == 0, "gemm_mx: scale arrays must be padded (mx_rows_pad)");
    ACE_CHECK((long)M * K < (1L << 32) && (long)N * K < (1L << 32), "gemm_mx: operands must each be smaller than 4 GB");
    GemmEpilogue ep = ep_in;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int per16 = (ep.mode == 0 || ep.mode == 3 || ep.mode == 4) ? 8 : 4;
    ep.wide_ok = al16(C) && (ldc % per16) == 0 && al16(ep.bias) && al16(ep.g1) && al16(ep.g2) && al16(ep.cvec) && (ep.g2_stride % 4) == 0;
    ACE_CHECK(ep.wide_ok, "gemm_mx: output / vectors must be 16-byte aligned");
    ACE_CHECK(al16(Aq) && al16(Wq) && al16(sa) && al16(sw), "gemm_mx: operands must be 16-byte aligned");
    ACE_CHECK(!ep.nf_xg && !ep.nc_rowsq, "gemm_mx: the folded-norm epilogue exists in the bf16 kernels only");
    {
        static int clk = -1;
        if (clk < 0) clk = env_int("ACE355_GEMM_CLK", 0);
        ep.clk_probe = clk;
    }
    ep.ksplit = 1;
    ep.kparts = 1;
    ep.sk_ord = 0;
    ep.vt_out = nullptr;
    if (ep.vt_done) *ep.vt_done = 0;
    ep.mx_sa = sa; ep.mx_sw = sw; ep.mx_sa_ld = sa_ld; ep.mx_sw_ld = sw_ld;
    if (ep.mode == 4)
        ACE_CHECK(ep.hn_wq && ep.hn_wk && (!ep.hn_cos == !ep.hn_sin) && ep.rows_per_seq > 0 && ep.hn_q_cols % 256 == 0 && ep.hn_qk_cols % 256 == 0 &&
                  ep.hn_qk_cols <= N && !ep.bias, "gemm_mx: head epilogue arguments");
    const int tiles_n = N / 256, tiles_m = (M + 191) / 192, nwg = tiles_m * tiles_n;
    // the kernel addresses rows in units of 2 bytes (its bf16 heritage): an fp8 row of K bytes is K / 2 such units, a K step of 128
    // fp8 elements is its 64-unit step
    const bf16_t* A = reinterpret_cast<const bf16_t*>(Aq);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(Wq);
    const int Kh = K / 2;
    switch (ep.mode) {
        case 0: launch_mx_mode<0>(s, A, Kh, W, Kh, C, ldc, M, N, Kh, ep, tiles_n, nwg); break;
        case 2: launch_mx_mode<2>(s, A, Kh, W, Kh, C, ldc, M, N, Kh, ep, tiles_n, nwg); break;
        case 3: launch_mx_mode<3>(s, A, Kh, W, Kh, C, ldc, M, N, Kh, ep, tiles_n, nwg); break;
        default: launch_mx_mode<4>(s, A, Kh, W, Kh, C, ldc, M, N, Kh, ep, tiles_n, nwg); break;
    }
    ACE_LAUNCH_CHECK();
    if (ep.clk_probe) {
        unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        ACE_HIP(hipStreamSynchronize(s));
        ACE_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk_probe), sizeof(h)));
        if (h[1]) fprintf(stderr, "[ace355 gemm-mx clk] M=%d N=%d K=%d mode=%d: %.3f GHz shader clock, %.0f cycles / K-step of 128 (%.3f us); prologue %.0f, "
                          "epilogue issue %.0f / acked %.0f cycles (last tile of workgroup 0)\n", M, N, K, ep.mode,
                          (double)h[0] / ((double)h[1] * 10.0), (double)h[0] / (double)h[2], (double)h[1] * 0.01 / (double)h[2],
                          (double)h[5], (double)h[3], (double)h[4]);
    }
    return 0;
}

}  // namespace ace355
