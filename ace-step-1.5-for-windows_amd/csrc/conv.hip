// conv.hip - Oobleck decoder convolutions as one MFMA implicit-GEMM kernel (channels-last, bf16, fp32 accumulate).
//
// Replaces, per call site of AutoencoderOobleck.decode (restated at acestep/models/mlx/vae_model.py):
//   * Snake1d -> Conv1d(k=7, dilated) and Snake1d -> Conv1d(k=1) + residual   (OobleckResidualUnit, :62-87)
//   * Snake1d -> ConvTranspose1d(k=2s, stride s, pad ceil(s/2))               (OobleckDecoderBlock, :119-142)
//   * the k=7 input conv and the Snake -> k=7 bias-free output conv           (OobleckDecoder, :190-230)
//
// One formulation covers all of them:  y[b, m, n] = bias[n] + sum_{tap, ci} f(x[b, m + (tap-center)*dil, ci]) * w[n][tap][ci]
// with f = Snake (x + 1/(e^beta+1e-9) * sin^2(e^alpha x)) or identity, zero rows outside [0, L_in).
// A stride-s transposed conv is the 2-tap case with N = s*Cout (all s polyphase filters side by side): row i0 of the
// GEMM is the contiguous NLC output span [(i0*s - pad)*Cout, ...+s*Cout), i.e. a flat shift of -pad*Cout (y_shift).
//
// gfx950 design: workgroup tile = 128 positions x BN outputs, K loop over 64-channel chunks; per chunk the input
// window (128 + (taps-1)*dil rows) is loaded ONCE, Snake applied in registers (fp32), and parked in LDS (swizzled);
// every tap then reads its shifted rows from LDS, so Snake costs (1 + halo) instead of `taps` evaluations and x is
// read from HBM/L2 once per chunk.  Weight tiles stream L2 -> LDS by DMA (global_load_lds) through a 2-stage ring, one barrier per tap
// (until round 3 they were staged global -> registers -> ds_write: 16 VGPRs and 4 LDS writes per thread per tap).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

// (the round-3 chunk / tap loop, the register-staged fused k = 1 stage, s_setprio in the tap loops and the de-phasing of the two resident
// workgroups were A/B arms of rounds 4-5: measured, not kept, removed in round 6 - git history up to dd0c588, numbers in DESIGN.md 13.7)

namespace ace355 {

namespace {

__device__ __forceinline__ int lds_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

// ACE355_CONV_CLK=1: shader-clock phases of one interior workgroup (thread 0): [0] total, [1] window staging + first weight
// tile (incl. the wait for the loads), [2] tap loops, [3] epilogue, [4] wall clock (100 MHz), [5] chunks.
__device__ unsigned long long g_conv_probe[8];

// Register-staged tiles use the native vector type: hipcc does not scalarise arrays of HIP's `uint4` struct that live across
// the tap loop (they ended up as LDS / scratch allocas: every weight prefetch was written out and waited for on the spot).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld_u32x4(const void* p) {
    return *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(reinterpret_cast<uintptr_t>(p));
}

// the same with a per-lane 64-bit address (the fused stage's parameter table: three separate vectors in one piece)
__device__ __forceinline__ void conv_glds16_v(const void* vptr, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(vptr), "s"(lds_addr)
                 : "memory");
}

constexpr int HALO_MAX = 64;  // (taps-1)*dil <= 6*9 = 54 rows of halo at most

// Weight tiles go L2 -> LDS by DMA (global_load_lds, issued from asm like the GEMM's: hipcc would drain a builtin DMA before the
// next ds_read): address = uniform 64-bit base (SGPR pair: the tap / channel-chunk offset) + a loop-invariant per-lane byte offset.
// A piece is 8 rows x 128 B, lane-linear in LDS; the XOR swizzle of lds_off is applied to the SOURCE chunk a lane fetches.
__device__ __forceinline__ void conv_glds16(unsigned voff, const void* sbase, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// K slots of the fused k = 1 stage.  The 8-wave form feeds snake2(t) to the MFMA straight from the k = 7 accumulators, and an accumulator
// lane (row, half h) holds channels 8 g + 4 h + (0..3) of every 32-channel tile: the eight values it can supply to K group q (16
// channels) are 16 q + 4 h + (0..3) and 16 q + 8 + 4 h + (0..3) - not the 8 h + (0..7) of a plain fragment.  Both forms of the stage
// therefore use THAT slot order for both operands (two 8-byte reads 16 bytes apart instead of one 16-byte read), so that a unit is the same
// bits whichever form launch_conv picks for its size (window decodes must equal whole-sequence decodes).
__device__ __forceinline__ bf16x8 frag_kperm(const char* base, int row, int q, int h) {
    const uint2 lo = *reinterpret_cast<const uint2*>(base + lds_off(row, 2 * q) + 8 * h);
    const uint2 hi = *reinterpret_cast<const uint2*>(base + lds_off(row, 2 * q + 1) + 8 * h);
    return as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// TM = positions per workgroup, one wave per 64 x 64 (BN = 128) sub-tile: 128 -> 4 waves, 256 -> 8 waves.  The 8-wave
// form shares each weight tile between twice as many MFMAs, shrinks the Snake halo from 1.42x to 1.21x of the tile and puts
// 4 waves on every SIMD (2 workgroups per CU either way), which is what hides the per-tap barrier and the staging phases.
// WS = 1 (TM = 256, BN = 128; the fused residual unit at C = 128 on the 8-wave tile, round 5): a wave owns 32 rows x ALL 128 output
// channels (MT = 1, NT = 4) instead of a 64 x 64 sub-tile, so the k = 7 result of its rows never has to meet another wave's: snake2(t)
// goes from the accumulators straight into the B operand of the k = 1 stage's MFMAs (no LDS image, no barrier between the stages).
template <int BN, int TM, int WS = 0>
__global__ __launch_bounds__(TM * 2, TM / 64) void conv_kernel(ConvArgs a) {
    static_assert(TM == 128 || (TM == 256 && BN == 128), "tile shapes");
    static_assert(WS == 0 || (TM == 256 && BN == 128), "the 32 x 128 wave shape belongs to the fused 8-wave form");
    constexpr int NTHR = TM * 2;
    constexpr int WIN_MAX = TM + HALO_MAX;
    constexpr int MT = WS ? 1 : ((BN == 128) ? 2 : 1);
    constexpr int NT = WS ? 4 : ((BN == 128) ? 2 : 1);
    constexpr int WCH = BN * 8 / NTHR;  // 16-B weight chunks per thread per tile
    // (4-wave 128 x 128 form: + a third 16 KB weight buffer and a 2 KB parameter table for the fused k = 1 stage: 74 KB, still two
    //  workgroups per CU like the 8-wave form's 72 KB; WS = 1: the table only, w2 goes through the ring)
    constexpr int FUSE_LDS = (BN == 128 && TM == 128) ? (BN * 128 + 2048) : (WS ? 2048 : 0);
    __shared__ __attribute__((aligned(16))) char smem[WIN_MAX * 128 + 2 * BN * 128 + FUSE_LDS];
    char* As = smem;
    char* Wbase = smem + WIN_MAX * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, half = lane >> 5;
    const int wm = WS ? wave : ((BN == 128) ? (wave >> 1) : wave);
    const int wn = WS ? 0 : ((BN == 128) ? (wave & 1) : 0);
    // Rasterisation.  Workgroups are handed to the 8 XCDs round-robin by linear id, each XCD has its own L2, and with the plain (m, n, b)
    // grid the column tiles of one row block are tiles_m launches-slots apart: every one of them pulled the same input window from HBM
    // (2x at C = 256 ... 8x at C = 1024 for the k = 1 convs, s * Cout / 128 times for the transposed ones).  With ras_tn > 1 the grid is
    // 1-D per batch item and ids 8q + r (r = XCD) walk the column tiles of row block (q / tn) * 8 + r on consecutive slots of XCD r:
    // they are resident together and share the window through that XCD's L2.  Same tiles, same arithmetic: results are bit-identical.
    int bx = blockIdx.x, by = blockIdx.y;
    if (a.ras_tn > 1) {  // workgroup-uniform
        const int L = blockIdx.x, tn = a.ras_tn, full = (a.ras_tm >> 3) << 3;
        if (L < full * tn) {
            const int q = L >> 3;
            by = q % tn;
            bx = (q / tn) * 8 + (L & 7);
        } else {  // the last (tiles_m % 8) row blocks: column tiles adjacent in time, XCDs as they come
            const int l2 = L - full * tn;
            by = l2 % tn;
            bx = full + l2 / tn;
        }
    }
    const int m0 = bx * TM, n0 = by * BN, b = blockIdx.z;
    const int Cin = a.Cin, taps = a.taps, dil = a.dil;
    const int win_rows = TM + (taps - 1) * dil;
    const int x_row0 = m0 - a.center * dil;
    const bf16_t* xb = a.x + (long)b * a.x_batch_stride;
    const long wrow = (long)taps * Cin;  // elements per output channel in w

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int sslot = tid & 7;
    // weight tile = BN rows x 128 B = BN / 8 DMA pieces; wave w issues pieces w, w + NW, ...: row = 8 piece + (lane >> 3), and the lane
    // fetches the logical 16-byte slot that lds_off puts at ITS physical position (slot' = lane & 7)
    constexpr int NWV = NTHR / 64;
    static_assert(WCH * NWV * 8 == BN, "pieces split evenly over the waves");
    unsigned w_voff[WCH];
    int wst[WCH];   // (register-staged form of the same tile: the fused k = 1 stage's w2 chunks)
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int prow = 8 * (wave + NWV * i) + (lane >> 3);
        const int ls = (lane & 7) ^ ((prow >> 1) & 7);
        w_voff[i] = (unsigned)(((long)min(n0 + prow, a.N - 1) * wrow + ls * 8) * 2);
        wst[i] = lds_off((tid >> 3) + (NTHR / 8) * i, sslot);
    }
    const unsigned wlds0 = (unsigned)(uintptr_t)Wbase + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    auto w_piece = [&](int i, int tap, int ci0, int buf) {   // piece i of tile (tap, chunk) -> weight buffer `buf`
        const bf16_t* sb = a.w + (long)tap * Cin + ci0;   // uniform
        const unsigned lb = (unsigned)__builtin_amdgcn_readfirstlane((int)(wlds0 + (unsigned)buf * (BN * 128)));
        conv_glds16(w_voff[i], sb, lb + (unsigned)i * (NWV * 1024));
    };
    auto w_issue = [&](int tap, int ci0, int buf) {
#pragma unroll
        for (int i = 0; i < WCH; ++i) w_piece(i, tap, ci0, buf);
    };
    // piece i of the 64-channel chunk c2 of w2 [128][128] (fused stage) -> the weight-tile image at `lbase` (this wave's first piece)
    auto w2_piece = [&](int i, int c2, unsigned lbase) {
        const int prow = 8 * (wave + NWV * i) + (lane >> 3);
        const int ls = (lane & 7) ^ ((prow >> 1) & 7);
        conv_glds16((unsigned)((prow * 128 + ls * 8) * 2), a.w2 + c2 * 64, (unsigned)__builtin_amdgcn_readfirstlane((int)lbase) + (unsigned)i * (NWV * 1024));
    };

    // Input window of one 64-channel chunk: all of a thread's (up to 6) 16-byte loads are issued back to back into
    // registers - and, for the next chunk, before the tap loop of the current one - so a workgroup pays one memory latency
    // per chunk instead of one per load (the loads used to sit in a load -> Snake -> ds_write loop).
    constexpr int WLD = (WIN_MAX * 8 + NTHR - 1) / NTHR;
    // the 4-wave form keeps the next chunk's window loads in flight across the tap loop; with 4 waves per SIMD the 8-wave
    // form has no registers to spare for that (128-VGPR budget) and other waves to cover the latency instead
    constexpr bool PREFETCH = true;   // (round 5: the 8-wave forms too - with descriptor loads the rows fit their 128 VGPRs across the taps)
    constexpr bool PARPRE = (TM == 128);                  // Snake parameters requested with the rows (16 more registers across the taps)
    u32x4 wv[WLD];
    const bool snake = a.alpha != nullptr;
    // V2 addressing: a buffer descriptor per workgroup and one loop-invariant 32-bit byte offset per thread (its row within a block of
    // NTHR / 8 window rows, its 16-byte slot) plus a uniform offset per (chunk, row block).  The hardware's range check replaces the
    // predicates: the descriptor covers [max(first window element, 0), end of the valid flat range) of this batch item, an offset
    // below it wraps to > 2^31 and one past it exceeds num_records - both read zeros, which is what rows outside the signal are
    // (rows beyond the window but inside the tensor are loaded and never stored).  Version 1 kept a 64-bit address and a lane mask
    // per load alive across the chunk loop: in the 8-wave form (128 VGPRs) they were spilled, and every reload's `s_waitcnt vmcnt(0)`
    // (scratch loads count in vmcnt) sat between two window loads - five serialised memory round trips per chunk instead of one.
    const int wrow_t = tid >> 3;                                        // row within a row block
    const unsigned x_voff = (unsigned)((wrow_t * Cin + sslot * 8) * 2);  // bytes; <= 64 rows x 2048 channels x 2 B
    const long xw_f0 = (long)x_row0 * Cin + a.x_shift;                   // flat index of (window row 0, channel 0); may be negative
    const long xw_end = a.x_valid ? a.x_valid : (long)a.L_in * Cin;      // the valid flat range of this item is [0, xw_end)
    const long xw_fb = max(xw_f0, 0L);
    const int xw_neg = (int)((xw_f0 - xw_fb) * 2);                        // <= 0: bytes from the descriptor's base back to window row 0
    const uintptr_t xw_bp = reinterpret_cast<uintptr_t>(xb + xw_fb);
    const unsigned xw_lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xw_bp);
    const unsigned xw_hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xw_bp >> 32));
    const unsigned xw_nrec = (unsigned)__builtin_amdgcn_readfirstlane((int)max(0L, min((xw_end - xw_fb) * 2, 0x7fffffffL)));
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)xw_hi32 << 32) | xw_lo32), (short)0, (int)xw_nrec, 0x00020000);
    auto win_load = [&](int ci0) {
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            // uniform part in the scalar offset (one address VGPR for all row blocks); where it is negative - row block 0 of a tile at the
            // start of the signal - it goes into the lane's offset instead, so that the range check sees the wrapped value
            const int uoff = xw_neg + (i * (NTHR / 8) * Cin + ci0) * 2;   // uniform
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)x_voff + min(uoff, 0), max(uoff, 0), 0);
            wv[i] = __builtin_bit_cast(u32x4, v);
        }
    };
    // Snake parameters of this thread's 8 channels of a chunk (exp(alpha), 1 / (exp(beta) + 1e-9): fp32 bits).  V2 requests them with
    // the window rows (ahead of the barrier that opens the chunk); they used to be loaded at the top of win_store, behind that
    // barrier, and waited for on the spot: one exposed L2 round trip per chunk
    u32x4 spa[2], spb[2];
    auto par_load = [&](int ci0) {
        if (snake) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                spa[q] = ld_u32x4(a.alpha + ci0 + sslot * 8 + 4 * q);
                spb[q] = ld_u32x4(a.beta + ci0 + sslot * 8 + 4 * q);
            }
        }
    };
    const int st0 = lds_off(tid >> 3, sslot);
    auto vec_rsrc = [&](const float* p, int n) {
        const uintptr_t u = reinterpret_cast<uintptr_t>(p);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uintptr_t)hi << 32) | lo), (short)0, p ? n * 4 : 0, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t sa_rsrc = vec_rsrc(a.alpha, Cin), sb_rsrc = vec_rsrc(a.beta, Cin);
    // Snake of one bf16 pair: x + 1/(e^beta + 1e-9) * sin^2(e^alpha x) (fp32; Snake(0) = 0, so zero rows - outside the signal or
    // beyond the window - pass through unchanged and the arithmetic needs no row predicate, only the store does)
    // (a0 / a1 arrive in REVOLUTIONS - exp(alpha) / 2 pi, folded once per thread and chunk by rev4 below: v_sin_f32 takes its argument
    //  in revolutions, and `__sinf(a * x)` spent a second multiply per element on the conversion)
    auto snake_pair = [&](unsigned p, float a0, unsigned b0, float a1, unsigned b1) -> unsigned {
        const float x0 = bf_lo(p), x1 = bf_hi(p);
        const float s0 = __builtin_amdgcn_sinf(a0 * x0), s1 = __builtin_amdgcn_sinf(a1 * x1);
        return pack_bf2(x0 + __uint_as_float(b0) * s0 * s0, x1 + __uint_as_float(b1) * s1 * s1);
    };
    auto rev4 = [&](const u32x4& v, float (&o)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __uint_as_float(v[e]) * 0.15915494309189535f;
    };
    auto win_store = [&](int ci0) {
        if (snake) {
            if constexpr (PARPRE) {   // parameters requested with the rows (spa / spb)
                float ra[2][4];
                rev4(spa[0], ra[0]); rev4(spa[1], ra[1]);
#pragma unroll
                for (int i = 0; i < WLD; ++i)
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2)
                        wv[i][e2] = snake_pair(wv[i][e2], ra[e2 >> 1][2 * (e2 & 1)], spb[e2 >> 1][2 * (e2 & 1)],
                                               ra[e2 >> 1][2 * (e2 & 1) + 1], spb[e2 >> 1][2 * (e2 & 1) + 1]);
            } else {
                // 8-wave form (128 VGPRs): the parameters of four channels at a time; with all sixteen values requested at once hipcc
                // spilled them as they arrived (load, vmcnt(0), scratch store, next load)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    // (descriptor + 32-bit lane offset + scalar offset: no 64-bit address VGPRs to keep - or spill - across the chunk loop)
                    const u32x4 pa = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sa_rsrc, sslot * 32, (ci0 + 4 * h) * 4, 0));
                    const u32x4 pb = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(sb_rsrc, sslot * 32, (ci0 + 4 * h) * 4, 0));
                    float ra[4];
                    rev4(pa, ra);
#pragma unroll
                    for (int i = 0; i < WLD; ++i)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            wv[i][2 * h + q] = snake_pair(wv[i][2 * h + q], ra[2 * q], pb[2 * q], ra[2 * q + 1], pb[2 * q + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int c = tid + i * NTHR;
            // row block i is NTHR / 8 = 32 or 64 rows further down: (row >> 1) & 7 - the swizzle key - does not change, so the
            // address is the thread's block-0 address plus an immediate (hipcc kept one address VGPR per block)
            if (c < win_rows * 8) *reinterpret_cast<u32x4*>(As + st0 + i * (NTHR / 8) * 128) = wv[i];
        }
    };
    // (wave-uniform condition: the clock values stay in SGPRs; with a per-thread `tid == 0` they lived in ten VGPRs of every wave)
    const bool probe = a.clk_probe && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == 0 && __builtin_amdgcn_readfirstlane(wave) == 0;
    unsigned long long p_t0 = 0, p_w0 = 0, p_stage = 0, p_taps = 0, p_mark = 0;
    if (probe) p_t0 = clock64(), p_w0 = wall_clock64();
    // Version 2 of the chunk / tap loop (round 5; same MFMA order per output element: results are bit-identical to version 1).
    // What the ISA of version 1 showed (hipcc --save-temps): (a) the Snake parameter loads of a chunk sat behind the chunk's first
    // barrier and were waited for at once; their destination registers were then reused for the tap loop's fragments, and because
    // win_store - wait included - is skipped by waves without window rows, hipcc guarded those registers with `s_waitcnt vmcnt(2)` /
    // `vmcnt(0)` at the top of EVERY tap: right behind the asm-issued DMA of the next tap's weight tile (invisible to hipcc), i.e.
    // every tap waited for the tile it had just requested and the DMA never ran under the MFMAs (a tap cost ~1530 cycles for 512
    // cycles of MFMA issue, and a deeper ring changed nothing); (b) the next chunk's window rows were requested ahead of that same
    // wait; (c) every fragment read was followed by `lgkmcnt(0)` and its MFMAs.  Here: parameters travel with the window rows, a
    // BUILTIN vmcnt(0) (visible to hipcc's counter model) closes the staging on every path, the tap-0 tile is requested ahead of the
    // Snake arithmetic, the next chunk's window rows behind the staging barrier (they land under tap 0), and the fragments of K
    // group kk + 1 are requested ahead of the MFMAs of group kk (4-wave form: registers to spare).
    constexpr bool FPIPE = (TM == 128);   // (8-wave forms: no registers for a second fragment set - the fused form tried it and spilled 17)
    if (PREFETCH) { win_load(0); if (PARPRE) par_load(0); }
    for (int ci0 = 0; ci0 < Cin; ci0 += 64) {
        if (probe) p_mark = clock64();
        if (!PREFETCH) win_load(ci0);
        __syncthreads();  // previous chunk fully consumed (window rows and both weight buffers)
        // vmcnt(0) as a BUILTIN: hipcc's counter model sees it, so win_store carries no waits of its own (they would also wait for the
        // DMA below, which hipcc does not track) and no register is guarded against these loads later.  The rows and parameters were
        // requested a tap loop ago (4-wave form) / ahead of the barrier (8-wave form).
        if (PREFETCH) __builtin_amdgcn_s_waitcnt(0x0F70);
        w_issue(0, ci0, 0);   // the tap-0 tile lands under the Snake arithmetic (8-wave form: with the parameters, one round trip for both)
        win_store(ci0);       // Snake in fp32 registers, parked in LDS once per chunk
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA
        __syncthreads();
        if (PREFETCH && ci0 + 64 < Cin) { win_load(ci0 + 64); if (PARPRE) par_load(ci0 + 64); }
        if (probe) { const unsigned long long t = clock64(); p_stage += t - p_mark; p_mark = t; }

#pragma unroll 1   // (hipcc unrolled the WS = 1 form's tap loop by its trip-count guess and hoisted every tap's fragment addresses: 169 VGPRs)
        for (int tap = 0; tap < taps; ++tap) {
            const bool more = (tap + 1) < taps;
            if (more) w_issue(tap + 1, ci0, (tap + 1) & 1);
            if constexpr (FUSE_LDS != 0) {
                // fused residual unit, last chunk: the k = 1 stage's operands arrive under the taps.  4-wave form: w2's first 64-channel
                // chunk one piece per wave per tap (taps 1-4) into the third buffer, w2's second chunk at the last tap into the ring buffer
                // that tap leaves free; 8-wave form (no third buffer): the FIRST chunk at the last tap into the free ring buffer, the second
                // behind the barrier that closes the taps.  The parameter table (bias | snake2 alpha | snake2 beta | bias2: 4 x 512 B) at tap
                // 5, one piece from each of waves 0 and 1 (launch_conv: taps == 7)
                if (a.w2 && ci0 + 64 >= Cin) {   // workgroup-uniform
                    constexpr int PT_OFF = (WS ? 2 : 3) * BN * 128;
                    if constexpr (WS == 0) {
                        const unsigned w2b = (unsigned)(uintptr_t)(Wbase + 2 * BN * 128) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
                        if (tap >= 1 && tap <= WCH) w2_piece(tap - 1, 0, w2b);
                    }
                    if (tap == 5 && __builtin_amdgcn_readfirstlane(wave) < 2) {
                        const float* src = (wave == 0) ? (lane < 32 ? (a.bias ? a.bias : a.alpha2) : a.alpha2)
                                                       : (lane < 32 ? a.beta2 : (a.bias2 ? a.bias2 : a.beta2));
                        conv_glds16_v(src + 4 * (lane & 31), (unsigned)(uintptr_t)(Wbase + PT_OFF) + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u);
                    }
                    if (!more) {
                        const unsigned ring = wlds0 + (unsigned)((tap + 1) & 1) * (BN * 128);
#pragma unroll
                        for (int i = 0; i < WCH; ++i) w2_piece(i, WS ? 0 : 1, ring);
                    }
                }
            }
            const char* Ws = Wbase + (tap & 1) * (BN * 128);
            const int arow = wm * (MT * 32) + tap * dil + lq;
            if constexpr (FPIPE) {
                bf16x8 fa[2][MT], fw[2][NT];
                auto frag = [&](int kk, int sl) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        fa[sl][i] = as_bf16x8(*reinterpret_cast<const uint4*>(As + lds_off(arow + i * 32, kk * 2 + half)));
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        fw[sl][j] = as_bf16x8(*reinterpret_cast<const uint4*>(Ws + lds_off(wn * (NT * 32) + j * 32 + lq, kk * 2 + half)));
                };
                frag(0, 0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kk < 3) frag(kk + 1, (kk + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(fw[kk & 1][j], fa[kk & 1][i], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    bf16x8 fa[MT], fw[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        fa[i] = as_bf16x8(*reinterpret_cast<const uint4*>(As + lds_off(arow + i * 32, kk * 2 + half)));
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        fw[j] = as_bf16x8(*reinterpret_cast<const uint4*>(Ws + lds_off(wn * (NT * 32) + j * 32 + lq, kk * 2 + half)));
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(fw[j], fa[i], acc[i][j]);
                }
            }
            if (more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        if (probe) p_taps += clock64() - p_mark;
    }
    // ---- epilogue operands (declared ahead of the fused stage, which requests them under its MFMAs)
    const bool full = (m0 + TM <= a.M) && (n0 + BN <= a.N) &&
                      ((long)m0 * a.N + n0 + a.y_shift >= 0) && ((long)(m0 + TM - 1) * a.N + n0 + BN - 1 + a.y_shift < a.y_valid);
    const bool wide = BN == 128 && a.out_mode == 0 && full && a.wide_ok;   // the LDS-staged 16-byte epilogue
    const int mw0 = m0 + wm * (MT * 32), nw0 = n0 + wn * (NT * 32);
    const int row_l = lane >> 2, c4 = lane & 3;
    u32x4 bl[NT], bh[NT];   // bias of this lane's 8 columns per half (fp32 bits)
    u32x4 rv[2][MT * 2];    // residual rows, double-buffered over the halves
    // (the 8-wave form is never launched with a residual: launch_conv; its 128-VGPR budget has no room for the residual rows)
    const bool has_res = (TM == 128 || WS == 1) && a.res != nullptr;
    auto res_load = [&](int j, u32x4 (&r)[MT * 2]) {
        const long rcol = (long)b * a.res_batch_stride + nw0 + j * 32 + c4 * 8 + a.y_shift;
#pragma unroll
        for (int t = 0; t < MT * 2; ++t) r[t] = ld_u32x4(a.res + rcol + (long)(mw0 + t * 16 + row_l) * a.N);
    };
    auto epi_pre = [&](const float* eb) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            bl[j] = u32x4{0u, 0u, 0u, 0u};
            bh[j] = u32x4{0u, 0u, 0u, 0u};
        }
        if (eb && WS == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bl[j] = ld_u32x4(eb + nw0 + j * 32 + c4 * 8);
                bh[j] = ld_u32x4(eb + nw0 + j * 32 + c4 * 8 + 4);
            }
        }
        if (has_res) res_load(0, rv[0]);
    };
    bool epi_pre_done = false;
    // ---------------------------------------------------------------- fused k = 1 stage of a residual unit (C = 128)
    // The k = 7 result t = acc + bias never leaves the workgroup: snake2(t) is written to LDS as the A operand of a
    // 128 x 128 x 128 GEMM with w2 (two 64-channel planes over the dead window / first weight buffer, w2 chunks through the
    // second weight buffer), and the epilogue below then adds bias2 and the residual.  Two of the unit's five tensor passes
    // (write t, read t) and one launch disappear.
    const float* epi_bias = a.bias;
    bool bias_in_lds = false;   // WS = 1 fused: bias2 is read from the parameter table (no registers for it across the epilogue)
    if constexpr (WS == 1) {
        if (a.w2) {  // workgroup-uniform
            epi_bias = a.bias2;
            bias_in_lds = a.bias2 != nullptr;
            const float b7s = a.bias ? 1.f : 0.f;
            const char* W2c0 = Wbase + (taps & 1) * (BN * 128);        // w2 channels 0-63: the ring buffer the last tap left free
            const char* W2c1 = Wbase + ((taps + 1) & 1) * (BN * 128);  // w2 channels 64-127: the last tap's own buffer, requested below
            const char* Pt = Wbase + 2 * BN * 128;                     // bias | alpha2 | beta2 | bias2 (landed under tap 5)
            __syncthreads();  // every wave is done with the window and the last tap's weight tile
            {
                const unsigned ring = wlds0 + (unsigned)(taps & 1 ? 0 : 1) * (BN * 128);
#pragma unroll
                for (int i = 0; i < WCH; ++i) w2_piece(i, 1, ring);   // lands under the Snake arithmetic
            }
            // snake2(k7 + bias) of this wave's 32 rows x 128 channels, packed to bf16 in MFMA B-operand order: sp[j][g] = channels
            // 32 j + 8 g + 4 half + (0..3) of row lq
            uint2 sp[NT][4];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {   // one register quad at a time: 12 parameter registers live, 4 accumulators become 2 packed ones
                    const int c = j * 32 + 8 * g + 4 * half;
                    const u32x4 b4 = *reinterpret_cast<const u32x4*>(Pt + c * 4);
                    const u32x4 e4 = *reinterpret_cast<const u32x4*>(Pt + 512 + c * 4);
                    const u32x4 i4 = *reinterpret_cast<const u32x4*>(Pt + 1024 + c * 4);
                    float t[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[0][j][4 * g + e] + b7s * __uint_as_float(b4[e]);
                        const float sn = __sinf(__uint_as_float(e4[e]) * v);
                        t[e] = v + __uint_as_float(i4[e]) * sn * sn;
                    }
                    sp[j][g] = make_uint2(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of both w2 chunks
            __syncthreads();
            // 128 x 128 K per wave row block: K group kg = channels 16 kg .. 16 kg + 15 (slot order: frag_kperm), B operand from sp
#pragma unroll
            for (int kg = 0; kg < 8; ++kg) {
                const char* Wp = (kg >> 2) ? W2c1 : W2c0;
                const uint2 s0 = sp[kg >> 1][2 * (kg & 1)], s1 = sp[kg >> 1][2 * (kg & 1) + 1];
                const bf16x8 fb = as_bf16x8(make_uint4(s0.x, s0.y, s1.x, s1.y));
                bf16x8 fw[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) fw[j] = frag_kperm(Wp, j * 32 + lq, kg & 3, half);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[0][j] = mfma32(fw[j], fb, acc[0][j]);
            }
        }
    }
    if constexpr (BN == 128 && TM == 128) {
        if (a.w2) {  // workgroup-uniform
            epi_bias = a.bias2;
            const float* b7p = a.bias ? a.bias : a.alpha2;  // (unconditional loads: a branch per element would fence them)
            const float b7s = a.bias ? 1.f : 0.f;
            __syncthreads();  // every wave is done with the window and the weight tiles
            char* A2 = smem;                      // 2 planes x 128 rows x 128 B (over the window and the head of ring buffer 0)
            const char* W2c0 = Wbase + 2 * BN * 128;             // w2 channels 0-63: the third buffer (landed under taps 1-4)
            const char* W2c1 = Wbase + (taps & 1) * (BN * 128);  // w2 channels 64-127: ring buffer 1 (requested at the last tap)
            const char* Pt = Wbase + 3 * BN * 128;               // bias | alpha2 | beta2 | bias2, 128 floats each (landed under tap 5)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 b4[4], e4[4], i4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = n0 + wn * (NT * 32) + j * 32 + 8 * g + 4 * half;
                    b4[g] = *reinterpret_cast<const u32x4*>(Pt + c * 4);
                    e4[g] = *reinterpret_cast<const u32x4*>(Pt + 512 + c * 4);
                    i4[g] = *reinterpret_cast<const u32x4*>(Pt + 1024 + c * 4);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float t[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[i][j][4 * g + e] + b7s * __uint_as_float(b4[g][e]);
                            const float sn = __sinf(__uint_as_float(e4[g][e]) * v);
                            t[e] = v + __uint_as_float(i4[g][e]) * sn * sn;
                        }
                        const int row = wm * (MT * 32) + i * 32 + lq;
                        uint2 pk = make_uint2(pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3]));
                        *reinterpret_cast<uint2*>(A2 + wn * 16384 + lds_off(row, j * 4 + g) + 8 * half) = pk;
                    }
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of w2's second chunk
            __syncthreads();
            if (wide) { epi_pre(a.bias2); epi_pre_done = true; }   // bias2 + the residual rows arrive under the 32 MFMAs below
            {   // 128 x 128 x 128: 8 K groups straight through, fragments one group ahead
                bf16x8 fa[2][MT], fw[2][NT];
                auto frag2 = [&](int k8, int sl) {
                    const char* Ap = A2 + (k8 >> 2) * 16384;
                    const char* Wp = (k8 >> 2) ? W2c1 : W2c0;
                    const int kk = k8 & 3;
#pragma unroll
                    for (int i = 0; i < MT; ++i) fa[sl][i] = frag_kperm(Ap, wm * (MT * 32) + i * 32 + lq, kk, half);
#pragma unroll
                    for (int j = 0; j < NT; ++j) fw[sl][j] = frag_kperm(Wp, wn * (NT * 32) + j * 32 + lq, kk, half);
                };
                frag2(0, 0);
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) {
                    if (k8 < 7) frag2(k8 + 1, (k8 + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = mfma32(fw[k8 & 1][j], fa[k8 & 1][i], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    const unsigned long long p_e0 = probe ? clock64() : 0ull;
    auto probe_done = [&]() {
        if (!probe) return;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // stores acknowledged
        const unsigned long long t = clock64();
        g_conv_probe[0] = t - p_t0, g_conv_probe[1] = p_stage, g_conv_probe[2] = p_taps, g_conv_probe[3] = t - p_e0;
        g_conv_probe[4] = wall_clock64() - p_w0, g_conv_probe[5] = (unsigned long long)(Cin / 64);
    };

    // ---------------------------------------------------------------- epilogue
    // Operands are swapped in the MFMAs, so lane l holds output ROW (position) lq of each 32x32 tile and, per register quad
    // g = r>>2, four consecutive channels n = .. + 8g + 4*half + (r&3).  Stores (and the residual loads) of an MFMA epilogue
    // are issue-bound per instruction: interior bf16 tiles go through a wave-private fp32 LDS staging image, one 32-channel
    // half at a time, and leave as 16-byte row-major accesses (8 loads + 8 stores per lane instead of 64 + 64 two-byte ones).
    if (wide) {  // workgroup-uniform
        // Every global load of the epilogue is requested before it is needed: the bias of both column halves and the first
        // half's residual rows before the staging (fused unit, V2: before the k = 1 stage's MFMAs), the second half's residual rows
        // before the first half is written.  (They used to be issued per half, each behind its own wait - eight scalar bias loads
        // behind eight branches, then the residual rows: four L2 round trips per workgroup, most of the 5.5-7 k cycle epilogue.)
        if (!epi_pre_done) epi_pre(epi_bias);
        __syncthreads();  // every wave is done with the operand tiles: the LDS may be overwritten
        char* stg = smem + wave * (MT * 32 * 128);  // MT*32 rows x 128 B (32 floats)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v;
                    v.x = acc[i][j][4 * g + 0]; v.y = acc[i][j][4 * g + 1]; v.z = acc[i][j][4 * g + 2]; v.w = acc[i][j][4 * g + 3];
                    const int row = i * 32 + lq, slot = 2 * g + half;
                    *reinterpret_cast<float4*>(stg + row * 128 + ((slot ^ (row & 7)) << 4)) = v;
                }
            if (has_res && j + 1 < NT) res_load(j + 1, rv[(j + 1) & 1]);
            u32x4 oa[2], ob[2];   // the consumer's Snake parameters of this lane's 8 columns (fp32 bits)
            if (a.osnake_a) {     // workgroup-uniform
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    oa[q] = ld_u32x4(a.osnake_a + nw0 + j * 32 + c4 * 8 + 4 * q);
                    ob[q] = ld_u32x4(a.osnake_b + nw0 + j * 32 + c4 * 8 + 4 * q);
                }
            }
            const long col = (long)b * a.y_batch_stride + nw0 + j * 32 + c4 * 8 + a.y_shift;
            if constexpr (WS == 1) {   // 128-VGPR form: this half's bias from the parameter table in LDS (fused unit), else from memory
                if (bias_in_lds) {
                    const char* pb = Wbase + 2 * BN * 128 + 1536 + (nw0 + j * 32 + c4 * 8) * 4;
                    bl[j] = *reinterpret_cast<const u32x4*>(pb);
                    bh[j] = *reinterpret_cast<const u32x4*>(pb + 16);
                } else if (epi_bias) {
                    bl[j] = ld_u32x4(epi_bias + nw0 + j * 32 + c4 * 8);
                    bh[j] = ld_u32x4(epi_bias + nw0 + j * 32 + c4 * 8 + 4);
                }
            }
            const float bias[8] = {__uint_as_float(bl[j][0]), __uint_as_float(bl[j][1]), __uint_as_float(bl[j][2]), __uint_as_float(bl[j][3]),
                                   __uint_as_float(bh[j][0]), __uint_as_float(bh[j][1]), __uint_as_float(bh[j][2]), __uint_as_float(bh[j][3])};
#pragma unroll
            for (int t = 0; t < MT * 2; ++t) {
                const int row = t * 16 + row_l;
                const float4 lo = *reinterpret_cast<const float4*>(stg + row * 128 + (((2 * c4) ^ (row & 7)) << 4));
                const float4 hi = *reinterpret_cast<const float4*>(stg + row * 128 + (((2 * c4 + 1) ^ (row & 7)) << 4));
                float v[8] = {lo.x + bias[0], lo.y + bias[1], lo.z + bias[2], lo.w + bias[3],
                              hi.x + bias[4], hi.y + bias[5], hi.z + bias[6], hi.w + bias[7]};
                if (has_res) {
                    const u32x4 r = rv[j & 1][t];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) { v[2 * e2] += bf_lo(r[e2]); v[2 * e2 + 1] += bf_hi(r[e2]); }
                }
                if (a.osnake_a) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sn = __sinf(__uint_as_float(oa[e >> 2][e & 3]) * v[e]);
                        v[e] = v[e] + __uint_as_float(ob[e >> 2][e & 3]) * sn * sn;
                    }
                }
                u32x4 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(a.y) + col + (long)(mw0 + row) * a.N) = pk;
            }
        }
        probe_done();
        return;
    }
    // generic per-element path (edge tiles, cropped transposed-conv spans, the fp32 NCL output of the last conv)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = mw0 + i * 32 + lq;
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = nw0 + j * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                if (n >= a.N) continue;
                float v = acc[i][j][r] + (epi_bias ? epi_bias[n] : 0.f);
                if (a.out_mode == 0) {
                    const long flat = (long)m * a.N + n + a.y_shift;
                    if (flat < 0 || flat >= a.y_valid) continue;
                    if (a.res) v += bf2f(a.res[(long)b * a.res_batch_stride + flat]);
                    if (a.osnake_a) {
                        const float sn = __sinf(a.osnake_a[n] * v);
                        v = v + a.osnake_b[n] * sn * sn;
                    }
                    reinterpret_cast<bf16_t*>(a.y)[(long)b * a.y_batch_stride + flat] = f2bf(v);
                } else {  // f32 NCL [b][n][m]
                    if (n < a.n_real) {
                        if (a.ncl_ld == 0) reinterpret_cast<float*>(a.y)[(long)b * a.y_batch_stride + (long)n * a.M + m] = v;
                        else if (m >= a.ncl_m_lo && m < a.ncl_m_hi)
                            reinterpret_cast<float*>(a.y)[(long)b * a.y_batch_stride + (long)n * a.ncl_ld + a.ncl_off + (m - a.ncl_m_lo)] = v;
                    }
                }
            }
    }
}

// z f32 [B][C][T] (NCL) -> bf16 [B][T][C] (NLC)
__global__ void ncl_to_nlc_kernel(const float* __restrict__ z, bf16_t* __restrict__ out, int C, int T, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const long bt = i / C;
    const int t = (int)(bt % T);
    const long b = bt / T;
    out[i] = f2bf(z[(b * C + c) * T + t]);
}

}  // namespace

int launch_conv(const ConvArgs& a, hipStream_t s) {
    ACE_CHECK(a.Cin % 64 == 0, "conv: Cin must be a multiple of 64");
    ACE_CHECK(a.taps >= 1 && (a.taps - 1) * a.dil <= HALO_MAX && a.dil >= 1, "conv: window too large");
    ACE_CHECK(a.B > 0 && a.M > 0 && a.N > 0, "conv: empty problem");
    ACE_CHECK((long)a.N * a.taps * a.Cin * 2 < (1L << 32) && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0,
              "conv: the weight tensor must be 16-byte aligned and smaller than 4 GB (32-bit DMA offsets)");
    ACE_CHECK(a.x_valid ? (a.x_shift % 8 == 0 && a.x_valid % 8 == 0) : a.x_shift == 0, "conv: x_shift / x_valid must be multiples of 8 (and x_shift needs x_valid)");
    ACE_CHECK(!a.w2 || (a.Cin == 128 && a.N == 128 && a.taps == 7 && !a.x_valid && a.out_mode == 0 && a.alpha2 && a.beta2 &&
                        (reinterpret_cast<uintptr_t>(a.alpha2) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.beta2) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.w2) & 15) == 0),
              "conv: the fused k = 1 stage needs Cin = N = 128, a plain conv and 16-byte aligned vectors");
    ACE_CHECK(!a.osnake_a == !a.osnake_b && (!a.osnake_a || a.out_mode == 0), "conv: the producer-side Snake needs both parameter vectors and the bf16 output");
    ConvArgs aw = a;
    {
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        aw.wide_ok = a.out_mode == 0 && al16(a.y) && al16(a.res) && al16(a.bias) && al16(a.bias2) && al16(a.osnake_a) && al16(a.osnake_b) &&
                     (a.N % 8) == 0 && (a.y_shift % 8) == 0 &&
                     (a.y_batch_stride % 8) == 0 && (a.res_batch_stride % 8) == 0;
    }
    static int tm_env = -1, clk_env = -1, ras_env = -1;
    if (ras_env < 0) {
        const char* e = getenv("ACE355_CONV_RAS");  // 0: the plain (m, n, b) grid (A/B runs); default 1: XCD-aware rasterisation
        ras_env = e ? atoi(e) : 1;
    }
    if (clk_env < 0) {
        const char* e = getenv("ACE355_CONV_CLK");
        clk_env = e ? atoi(e) : 0;
    }
    aw.clk_probe = clk_env;
    if (tm_env < 0) {
        const char* e = getenv("ACE355_CONV_TM");  // 128 / 256 force a tile height (A/B runs); default: by problem size
        tm_env = e ? atoi(e) : 0;
    }
    if (a.N >= 128 || a.N % 128 == 0) {
        const long wgs128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128) * a.B;
        // 8-wave tiles pay off where the tap loop dominates (Cin >= 256: +2-5 %); at Cin = 128 and for k = 1 the exposed
        // window load of the un-prefetched form costs more than the sharing saves (-5 % / -35 %)
        // (the 2-tap transposed / strided convs are faster on the 4-wave tile at every stage: 1.31 / 1.34 / 1.46 ms against
        //  1.44 / 1.50 / 1.83 ms)
        // (the fused k = 1 stage of a residual unit is written for the 4-wave 128-row tile only: forcing the tall tile onto it with
        //  ACE355_CONV_TM=256 gave a - 6 dB decode, caught by test_decode_at_the_metric_length_vs_oracle; it is refused here)
        // Round 4, after the Snake move (the transposed convs stage plain rows now) and a per-launch sweep of the forced heights
        // (profiles/r04/r04_conv_tile_height_sweep.txt, 8 x 30 s): the k = 7 convs at C = 1024 are 3776 four-wave tiles - below the old 4096
        // threshold - and run 1076-1097 us on them against 875-897 us on 1888 tall ones: (threshold lowered, see below); the plain-row transposed convs with Cin >= 1024 gain too (782 -> 689 and 1096 -> 1008 us), the ones with
        // Cin <= 512 lose (1128 -> 1174, 1349 -> 1510, 1660 -> 2112 us) and stay on the four-wave tile, like the k = 1 convs (+ 5-25 %).
        // The same sweep at 4 / 2 / 1 songs (profiles/r04/r04_conv_tile_height_sweep_small_batches.txt) shows the size threshold itself was
        // wrong for the k = 7 convs: the tall tile wins at every batch (one song: 205 -> 190, 259 -> 221, 240 -> 197 us at C = 1024 / 512 /
        // 256; two songs: 328 -> 254, 450 -> 376 us), down to the 472 four-wave tiles of one 30 s song at C = 1024: threshold 448.  The
        // plain-row transposed convs with Cin >= 1024 gain from ~ 900 four-wave tiles up (one song, 2048 -> 1024, 480 tiles: 149 -> 170 us).
        // Round 5, after version 2 of the chunk / tap loop (the 8-wave form no longer spills, its window loads are no longer serialised
        // by scratch reloads): re-swept at 1 / 2 / 4 / 8 songs (profiles/r05/r05_conv_tile_height_sweep.txt).  The 8-wave tile now wins on
        // EVERY k >= 2 launch that fills the chip about one and a half times (512 tall workgroups are resident at once = 1024 four-wave
        // tiles): the transposed convs at every Cin (8 songs: 1645 -> 1388, 1256 -> 1069, 1020 -> 845 us for Cin = 128 / 256 / 512), the k = 7
        // convs from ~1400 four-wave tiles (one song: C = 256 204 -> 175 us, C = 512 224 -> 221, C = 1024 (472 tiles) 177 -> 197: stays 4-wave;
        // two songs: all three win), the unfused k = 7 at C = 128 too (3.38 -> 2.98 ms).  Below that the tail of half-empty rounds costs more.
        const bool tall_k = a.Cin >= 128 && a.taps >= 3 && wgs128 >= 1400;
        const bool tall_t = a.taps == 2 && a.Cin >= 128 && !a.alpha && !a.x_valid && wgs128 >= 1500;
        const bool tall = !a.w2 && !a.res && (tm_env ? tm_env == 256 : (tall_k || tall_t));
        const int tm_rows = tall ? 256 : 128;
        dim3 grid((a.M + tm_rows - 1) / tm_rows, (a.N + 127) / 128, a.B);
        aw.ras_tm = aw.ras_tn = 0;
        if (ras_env && grid.y > 1 && (long)grid.x * grid.y < (1L << 31)) {
            aw.ras_tm = (int)grid.x, aw.ras_tn = (int)grid.y;
            grid = dim3(grid.x * grid.y, 1, a.B);
        }
        // fused residual unit (C = 128): the 8-wave form with 32 x 128 wave tiles from the same size threshold as the other k = 7 convs
        // (ACE355_CONV_F8=0 / 1: never / always, A/B runs); both forms give the same bits (frag_kperm)
        static int f8_env = -2;
        if (f8_env == -2) {
            const char* e = getenv("ACE355_CONV_F8");
            f8_env = e ? atoi(e) : -1;
        }
        const bool f8 = a.w2 && (f8_env >= 0 ? f8_env != 0 : wgs128 >= 1400);
        if (f8) {
            dim3 g8((a.M + 255) / 256, 1, a.B);
            aw.ras_tm = aw.ras_tn = 0;
            hipLaunchKernelGGL((conv_kernel<128, 256, 1>), g8, dim3(512), 0, s, aw);
        } else
        if (tall) hipLaunchKernelGGL((conv_kernel<128, 256>), grid, dim3(512), 0, s, aw);
        else hipLaunchKernelGGL((conv_kernel<128, 128>), grid, dim3(256), 0, s, aw);
    } else {
        aw.ras_tm = aw.ras_tn = 0;
        dim3 grid((a.M + 127) / 128, (a.N + 31) / 32, a.B);
        hipLaunchKernelGGL((conv_kernel<32, 128>), grid, dim3(256), 0, s, aw);
    }
    ACE_LAUNCH_CHECK();
    if (aw.clk_probe) {
        ACE_HIP(hipStreamSynchronize(s));
        unsigned long long h[8];
        ACE_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_conv_probe), sizeof(h)));
        fprintf(stderr, "[conv clk] Cin=%d N=%d taps=%d dil=%d M=%d: total %llu cyc = stage %llu + taps %llu + epilogue %llu (+%llu other), wall %.2f us, %llu chunks\n",
                a.Cin, a.N, a.taps, a.dil, a.M, h[0], h[1], h[2], h[3], h[0] - h[1] - h[2] - h[3], h[4] / 100.0, h[5]);
    }
    return 0;
}

int launch_ncl_to_nlc(const float* z, bf16_t* out, int B, int C, int T, hipStream_t s) {
    const long total = (long)B * C * T;
    hipLaunchKernelGGL(ncl_to_nlc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, out, C, T, total);
    ACE_LAUNCH_CHECK();
    return 0;
}

}  // namespace ace355
