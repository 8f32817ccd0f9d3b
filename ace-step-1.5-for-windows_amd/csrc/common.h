// common.h - shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

namespace ace355 {

// ------------------------------------------------------------------ error plumbing (host)
void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define ACE_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess) return ::ace355::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define ACE_CHECK(cond, msg)                                  \
    do {                                                      \
        if (!(cond)) {                                        \
            ::ace355::set_error(std::string("invalid argument: ") + (msg)); \
            return 1;                                         \
        }                                                     \
    } while (0)

#define ACE_LAUNCH_CHECK() ACE_HIP(hipGetLastError())

// ------------------------------------------------------------------ roctx ranges (the reference brackets its
// phases with profiler ranges, acestep/profile_inference.py).  ACE355_ROCTX=1 loads librocprofiler-sdk-roctx.so (or libroctx64.so) at first use (dlopen: no link-time
// dependency, nothing happens without the variable); rocprofv3 --marker-trace then shows sample / step / forward / layer / decode.
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char* name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// ------------------------------------------------------------------ device types
typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// round-to-nearest-even fp32 -> bf16 pair in one v_cvt_pk_bf16_f32 (gfx950)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// 16-byte load through the GLOBAL address space: a pointer that went through a select / table lookup is "flat" to the
// compiler, and flat loads also tick lgkmcnt, which would serialise them against every LDS wait.
__device__ __forceinline__ uint4 ldg16(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const u32x4_t v = *reinterpret_cast<const __attribute__((address_space(1))) u32x4_t*>(reinterpret_cast<uintptr_t>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// x * sigmoid(x) with the hardware exp2 / rcp (1 ulp; results are rounded to bf16 afterwards).  A plain `x / (1 + expf(-x))`
// compiles to the IEEE division sequence (~10 VALU ops per element): 5 % of the SwiGLU GEMM at 96 elements per lane.
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// MFMA 32x32x16 bf16: D[i][j] += sum_k A[i][k] B[k][j].
//   A operand: lane l holds A[i = l&31][k = (l>>5)*8 + e], e = 0..7
//   B operand: lane l holds B[k = (l>>5)*8 + e][j = l&31]
//   C/D:       lane l, reg r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31]
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------ host-side kernel launch API (internal)
struct GemmEpilogue {
    int mode;  // 0 store bf16 (+bias), 1 store f32 (+bias), 2 residual-gate into f32 H, 3 swiglu -> bf16 [M,N/2]
    const float* bias;
    const float* g1;
    const float* g2;
    int g2_stride;
    int rows_per_seq;
    const float* cvec;  // mode 2: rows m >= cvec_row0 additionally get + cvec[n] (constant cross-attention term of broadcast slots)
    int cvec_row0;
    int clk_probe;  // set by launch_gemm (ACE355_GEMM_CLK diagnostic)
    int ksplit;     // set by launch_gemm: > 1 = split-K (mode 2 only): blockIdx.y owns a K range and adds its share into H (see sk_cnt)
    int wide_ok;  // set by launch_gemm: C / ldc / per-column vectors are 16-byte aligned, so the 16-byte staged epilogue may be used
    // mode 4 (self-attention QKV projection with PACK_ROWS_HEADPAIR q / k rows): columns [0, hn_q_cols) are q heads, up to
    // hn_qk_cols k heads, the rest v.  q / k leave the GEMM head-normed (weights hn_wq / hn_wk [128], eps hn_eps) and rotated
    // (hn_cos / hn_sin [pos][64], pos = row % rows_per_seq) straight from the fp32 accumulators; v as in mode 0.  launch_gemm
    // falls back to mode 0 + headnorm_rope_kernel(paired) for tile shapes whose waves do not pair up over a head.
    const float *hn_wq, *hn_wk, *hn_cos, *hn_sin;
    int hn_q_cols, hn_qk_cols;
    float hn_eps;
    // MX-scaled fp8 operands (OCP MXFP8: e4m3 elements, one E8M0 scale per 32 consecutive K elements of a row; mx_quant_kernel).
    // Non-null: A and W point at fp8 bytes ([M, K] / [N, K] row-major, K % 128 == 0; lda / ldw in BYTES / 2, launch_gemm_mx does
    // that), the scales are uint32 [K / 128][rows_padded]: byte b of word [kt][r] = scale of row r, K elements 128 kt + 32 b .. + 31.
    const uint32_t* mx_sa; const uint32_t* mx_sw;
    int mx_sa_ld, mx_sw_ld;   // padded row counts of the two scale arrays (words per K step)
    // mode 3 only: the SwiGLU output leaves as MXFP8 (the down projection's operand) instead of bf16: C = fp8 [M, N/2] (ldc in bytes),
    // block scales to mxo_scales ([N/2 / 128][mxo_pad] words).  A wave's 32 output columns are exactly one block.
    uint32_t* mxo_scales; int mxo_pad;
    // RMSNorm folded into the neighbouring GEMMs (dit.hip forward_core, sampler path; base.py:493-533).  rmsnorm(h) * g + shift feeds a
    // projection W: (h * g) W^T * rstd_row + shift W^T.  Producer (mode 2, the GEMM that finishes h): besides H it writes
    // xg[m][n] = bf16(h_new * g[n]) and adds each row's sum of h_new^2 into rowsq[m]; rows m < nf_split use the A set (g vector, rowsq
    // array), the others the B set (the self-attention o_proj feeds the cross-attention norm on conditional rows and the MLP norm on the
    // rows that skip cross-attention).  Consumer (modes 0 / 3 / 4): acc = acc * rsqrt(rowsq[m] * nc_inv_d + nc_eps) + nc_bias[n]
    // (nc_bias = shift W^T, fp32 [N], NULL for an un-shifted norm) ahead of the mode's own epilogue.  Wide epilogue only (launch_gemm checks).
    bf16_t* nf_xg; int nf_ldx; int nf_split;
    const float* nf_gA; const float* nf_gB;
    // split-K (mode 2) in part order instead of fp32 atomics: the caller lends a zeroed counter array (SK_CNT_INTS ints, one stream at a
    // time); part y of a tile waits until parts 0 .. y-1 have added their share to H, then does the plain read-modify-write itself, so
    // the sum has one order and the result is bit-reproducible.  sk_ord is set by launch_gemm.  NULL: the atomics path.
    int* sk_cnt; int sk_ord;
    int kparts;   // set by launch_gemm: K parts over blockIdx.y (= ksplit; > 1 only for the ordered / atomic split-K of mode 2)
    // mode 4, v tiles (columns >= hn_qk_cols): written TRANSPOSED straight from the accumulators, vt[seq][head][d][s] (row pitch vt_ld,
    // s = row % rows_per_seq), instead of as rows of C - the separate transpose_v launch of a small-M forward disappears (2-byte stores,
    // 32 consecutive key positions per half wave: fine for the few tiles of a small-M launch, too slow for the big tiles, so launch_gemm
    // clears vt_out for those and reports through *vt_done what it did).  Pad positions [S, vt_ld) are the caller's to keep zero.
    bf16_t* vt_out; int vt_ld; int vt_heads; int* vt_done;
    unsigned long long* nf_sqA; unsigned long long* nf_sqB;   // 2^-24 fixed point: integer adds commute, so the sums (and with them
    const unsigned long long* nc_rowsq; const float* nc_bias;  // every result) do not depend on the order the atomics arrive in
    float nc_inv_d, nc_eps;
    int krot;   // kernel-visible: 1 = non-persistent bf16 launches walk K rotated by (workgroup's XCD) * nk / 8 (gemm.hip: K rotation)
    // cu_slots (0 = the whole chip, 256): the CUs this launch plans for - tile choice, persistent grid size, deep-pipeline decision.  The
    // dual-chain sampler (dit.hip) runs two independent launch sequences on two hardware queues and shapes every launch of both for 128
    // CUs, so that the chains run side by side instead of interleaving workgroups over all 256 (round 4: 512 -> 473 ms per 8-song pass)
    int cu_slots;
    // Weight prefetch for the NEXT GEMM of the caller's sequence (round 6, small requests; set by the caller through gemm_prefetch_plan, all
    // zero = none): a launch that leaves CUs without a tile carries `pf_x` extra workgroups per XCD that read, and throw away, the W rows the
    // next launch's workgroups ON THAT XCD will stream - [xj * pf_chunk16, + pf_chunk16) in 16-byte units, xj = the XCD's column region in the
    // next launch's XCD grid (8 / pf_xcd_n x pf_xcd_n) - so that they sit in that XCD's L2 when the next launch starts.
    const void* pf_w;
    unsigned pf_chunk16, pf_total16, pf_len16;   // (pf_len16 <= pf_chunk16: how much of a region is read - an L2 holds 4 MB)
    int pf_xcd_n, pf_x;
    // a_zero_idx > 0: row a_zero_idx of A (counted from the A pointer, >= M) exists and is all zeros: the rows a tile pads beyond M read IT
    // instead of repeating row M - 1.  The padded rows' MFMAs are wasted either way (M = 6000 on 192-row tiles: 2.4 % of all of them); on
    // zeros the matrix pipe switches next to nothing for them, and under the power cap that energy comes back as clock.
    int a_zero_idx;
    // a_wrap > 0: rows >= a_wrap of A read row (m - a_wrap): the operand of the second half of the rows IS the first half's (layer 0 of a
    // CFG forward: both halves carry the same latents up to the first cross-attention, dit.hip: forward_core).
    int a_wrap;
};
int gemm_set_k_rotation(int mode);   // ace355_gemm_set_k_rotation; returns the previous mode
int gemm_k_rotation_mode();          // the current mode (0: launch-shape-independent summation orders, also honoured by launch_attention)
// Fills ep->pf_* for a following launch_gemm(.., M, N, K, mode ..) whose weight matrix is W (row stride K): the XCD column regions that launch
// will use (same tile / XCD-grid choice as launch_gemm makes).  Leaves ep untouched when that launch would not be one the prefetch helps.
void gemm_prefetch_plan(GemmEpilogue* ep, const void* W, int M, int N, int K, int mode, int cu_slots);
int launch_gemm(const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M, int N, int K,
                const GemmEpilogue& ep, hipStream_t s);
// MXFP8 operands: Aq / Wq fp8 e4m3 [M, K] / [N, K] (row stride = K bytes), scales as GemmEpilogue::mx_sa / mx_sw describe; same
// epilogues (modes 0, 2, 3, 4).  Only shapes that take the 8-wave 192x256 tile (launch_gemm_mx checks); K % 128 == 0.
int launch_gemm_mx(const uint8_t* Aq, const uint32_t* sa, int sa_ld, const uint8_t* Wq, const uint32_t* sw, int sw_ld, void* C, int ldc,
                   int M, int N, int K, const GemmEpilogue& ep, hipStream_t s);
bool gemm_mx_supported(int M, int N, int K, int mode);
int gemm_verify_splitk_placement();   // one-time XCD placement check behind the small-M split-K (gemm.hip); handle constructors call it
bool gemm_fold_supported();   // the folded-RMSNorm epilogue hooks exist in the kernels launch_gemm will use (not the v1 bring-up kernel)
// x bf16 [M, K] (row stride ld) -> q fp8 e4m3 [M, K] + scales uint32 [K / 128][rows_pad] (rows_pad >= M, multiple of 4)
int launch_mx_quant(const bf16_t* x, long ld, int M, int K, uint8_t* q, uint32_t* scales, int rows_pad, hipStream_t s);
// rows of a packed bf16 weight matrix through e4m3 with one scale per row and back (fp8 weight-only semantics; elementwise.hip)
int launch_fp8_weight_roundtrip(bf16_t* w, long ld, int N, int K, hipStream_t s);
inline int mx_rows_pad(int rows) { return ((rows + 255) / 256) * 256 + 256; }

constexpr int SK_MAX_TILES = 4096;
constexpr int SK_CNT_INTS = SK_MAX_TILES + 8;   // size of a turn-counter array: the tiles' counters + [SK_MAX_TILES] = missed-turn count
// A part that waited ~1 s for its turn (a counter left behind by a faulted launch, or a placement the one-time probe did not see)
// goes ahead so the device cannot hang, but counts the event in sk_cnt[SK_MAX_TILES]: gemm_splitk_poll reads it (stream sync), and on a
// non-zero count re-zeroes the counters and reports an error - the residual stream of that call is not trustworthy.
int gemm_splitk_poll(int* sk_cnt, hipStream_t s);

struct AttnArgs {
    const bf16_t* q; long q_seq_stride; int q_row_stride;           // q[n][s][h*128 + d]
    const bf16_t* k; long k_seq_stride; long k_head_stride; int k_row_stride;  // k[n][hkv][s][d] general strides
    const bf16_t* vt; long vt_seq_stride; long vt_head_stride; int vt_ld;      // vt[n][hkv][d][s_pad]
    int use_tab;          // 1: per-sequence absolute base addresses below (cross-attention condition slots)
    unsigned long long k_tab[64];
    unsigned long long vt_tab[64];
    bf16_t* out; long o_seq_stride; int o_row_stride;
    int N, Sq, Skv, Hq, Hkv, window;
    float scale;
    int clk_probe;  // set by launch_attention (ACE355_ATTN_CLK diagnostic)
    // key-padding mask of the condition encoders (create_4d_mask with attention_mask, base.py:117-124), prefix form:
    // keys j >= kv_len[n] are masked.  A query row with NO valid key (padding query outside the band of the valid prefix)
    // softmaxes to uniform over ALL Skv keys in the reference (the mask is finfo.min, not -inf): it gets vmean[n][hkv][:].
    const int* kv_len;      // device [N] or null
    const bf16_t* vmean;    // device [N][Hkv][128], required when kv_len is set
    // MXFP8 output (the o_proj MX GEMM's operand, rows = n * Sq + s, Hq * 128 columns): out_q fp8 [N * Sq, Hq * 128], block scales
    // [Hq][out_pad] words (a head's 128 columns = one K step, byte dt = columns 32 dt .. 32 dt + 31).  attn_gqa_kernel only
    // (attention_mx_out_ok); `out` is not written then.
    uint8_t* out_q; uint32_t* out_scales; int out_pad;
    // split-KV for problems with too few workgroups to fill the chip (batch-1 requests: 48-96 workgroups walking 6-13 key tiles one
    // after the other): the caller lends a scratch (part, part_floats); launch_attention then lets `kv_split` workgroups share the key
    // tiles of one (q-block, head, sequence), each leaving un-normalised O (fp32), its running max and its row sum, and a second small
    // kernel merges them (same result as one pass up to fp32 rounding; order fixed, so bit-reproducible).  kv_split is set by launch_attention.
    float* part; long part_floats; int kv_split;
    int cu_slots;   // host-side hint, as GemmEpilogue::cu_slots: 0 = plan for the whole chip
};
int launch_attention(const AttnArgs& a, hipStream_t s);
bool attention_mx_out_ok(const AttnArgs& a);  // will launch_attention take the kernel that can write MXFP8?

struct ModEntry {  // one modulated norm: g = w * (1 + sc1 + tproj[sc2_off..]), sft = sh1 + tproj[sh2_off..]
    const float *w, *sc1, *sh1;
    long sc2_off, sh2_off;
};
int launch_rmsnorm_gs(const float* x, const float* g, const float* sft, bf16_t* y, int M, int D, float eps, long stride,
                      int rows_per_seq, hipStream_t s);
// the same norm with an MXFP8 output (q fp8 [M, D] + block scales): bit-identical to launch_rmsnorm_gs -> launch_mx_quant
int launch_rmsnorm_gs_mx(const float* x, const float* g, const float* sft, uint8_t* q, uint32_t* scales, int rows_pad, int M, int D,
                         float eps, long stride, int rows_per_seq, hipStream_t s);
int launch_mod_gs(const ModEntry* entries_dev, int n_entries, const float* tproj, long tp_stride, int rows, float* out, int D,
                  hipStream_t s);
// gs [rows][n_entries][2][D] (launch_mod_gs) -> the shift halves as bf16 rows, out [n_entries][rows][D]
int launch_shift_rows(const float* gs, int n_entries, int rows, bf16_t* out, int D, hipStream_t s);
int launch_rmsnorm_mod(const float* x, const float* w, bf16_t* y, int M, int D, float eps, const float* sc1,
                       const float* sc2, const float* sh1, const float* sh2, int stride, int rows_per_seq, hipStream_t s);
int launch_headnorm_rope(bf16_t* x, int M, int ld, int col0, int heads, const float* w, float eps,
                         const float* cos_tab, const float* sin_tab, int S, hipStream_t s);
// one launch for q heads [0,split) with weight w and k heads [split,heads) with weight w2 (contiguous columns)
int launch_headnorm_rope2(bf16_t* x, int M, int ld, int col0, int heads, const float* w, const float* w2, int split, float eps,
                          const float* cos_tab, const float* sin_tab, int S, hipStream_t s, int paired = 0);
// vt[n][h][d][s] (ld = s_pad) <- x[(n*S + s)*ld + col0 + h*128 + d]
int launch_transpose_v(const bf16_t* x, int ld, int col0, int N, int S, int heads, bf16_t* vt, int s_pad, hipStream_t s);
int launch_rope_table(float* cos_tab, float* sin_tab, int S, float theta, hipStream_t s);
// vmean[n][h][d] = mean over s in [0,S) of x[(n*S + s)*ld + col0 + h*128 + d]
int launch_vmean(const bf16_t* x, int ld, int col0, int N, int S, int heads, bf16_t* vmean, hipStream_t s);
// dst[r][c] (f32, row stride dst_ld) = src[r][c] (bf16, row stride src_ld) for `rows` rows taken through a per-row source index
// table (row_src[r] == -1: zero row, < -1: leave dst row r untouched)
// out[(r*P + p)][c] = emb[r][c] + special[p][c]   (f32)
int launch_expand_add(const float* emb, const float* special, float* out, long rows, int P, int D, hipStream_t s);
// audio tokenizer helpers (elementwise.hip): zero-padded bf16 rows, [special | P frames] sequences, token 0 of every sequence
int launch_pad_rows_bf16(const float* x, bf16_t* out, long rows, int cols, int ld, hipStream_t s);
int launch_prepend_special(const float* emb, const float* special, float* out, long rows, int P, int D, hipStream_t s);
int launch_take_token0(const bf16_t* x, float* out, long rows, int S, int D, hipStream_t s);
// out f32 [M, N] = x f32 [M, K] W^T (f32 [N, K]) + b (f32 [N] or null), fp64 accumulation: the FSQ projections of the LM-hint path
int launch_linear_f32(const float* x, const float* W, const float* b, float* out, long M, int N, int K, hipStream_t s);
int launch_gather_rows_bf16_f32(const bf16_t* src, long src_ld, const int* row_src, float* dst, long dst_ld, long rows, int cols, hipStream_t s);

struct TVals { float t[64]; };
int launch_sinusoid(const TVals& tv, int n, float* out /*[n,256]*/, hipStream_t s);
// out[m][n] (+)= sum_k act(in[m][k]) W[n][k] + b[n];  in f32 [Mr,K], W bf16 [N,K], out f32 [Mr,N]
int launch_small_linear(const float* in, const bf16_t* W, const float* b, float* out, int Mr, int N, int K,
                        int silu_in, int silu_out, int accumulate, hipStream_t s);
int launch_small_linear_ex(const float* in, const bf16_t* W, const float* b, float* out, float* out_silu, int Mr, int N,
                           int K, int silu_out, int accumulate, hipStream_t s);
// one wave that spins for `us` microseconds (wall clock): the stream-concurrency probe of the dual-chain sampler (dit.hip)
int launch_spin(int us, hipStream_t s);
int launch_pack_xin(const float* x, const float* ctx, bf16_t* xin, int N, int T, int Tpad, hipStream_t s);
int launch_set_xin_latent(const float* xt, bf16_t* xin, int B, int copies, int T, int Tpad, hipStream_t s);
int launch_set_xin_ctx(const float* ctx, bf16_t* xin, int B, int copies, int T, int Tpad, hipStream_t s);
int launch_f32_to_bf16(const float* in, bf16_t* out, long n, hipStream_t s);
int launch_expand_kv_heads(const bf16_t* v_row, bf16_t* out, int hq, int hkv, hipStream_t s);
int launch_bcast_rows(const bf16_t* in, bf16_t* out, int rows, int cols, hipStream_t s);
int launch_copy_v(const float* vpad, float* v, int N, int T, int Tpad, hipStream_t s);
struct StepUpdate {          // how x_t advances after the guided velocity is known
    const float* sde_noise;  // null: ODE Euler step x -= v*dt (base.py:1974-1979); else [B,T,64] fresh noise of this step:
    float t_curr, t_next;    //       x0 = x - v*t_curr; x = t_next*noise + (1-t_next)*x0 (base.py:1968-1973)
};
int launch_apg_euler(const float* v, long uncond_offset, float* avg, float* xt, bf16_t* xin, int copies, int B, int T,
                     int Tpad, float guidance, float dt, int apply_cfg, int do_cfg, int first, const StepUpdate& up, hipStream_t s);
int launch_adg_step(const float* v, long uncond_offset, float* xt, bf16_t* xin, int copies, int B, int T, int Tpad, float guidance,
                    float sigma, float dt, const StepUpdate& up, hipStream_t s);
int launch_peak_normalize(float* wav, int B, long per_item, float* scratch, hipStream_t s);
int launch_normalize_db(float* wav, int B, long per_item, float amp, float* peaks, hipStream_t s);
int launch_interleave(const float* wav, int B, int C, long S, void* out, int as_pcm16, hipStream_t s);
int launch_latent_check(const float* x, long n, int* flags_dev, hipStream_t s);

// generic pack: dst bf16 / f32 from src (f32 or bf16) with an index mapping
// PACK_ROWS_HEADPAIR: rows of a [heads*128, cols] projection land so that the rotate-half partners (d, d+64) of every head
// are neighbours: dst row = head*128 + (d < 64 ? 2d : 2(d-64)+1).  q.k is invariant under a common permutation of the head
// dims, and RoPE / head-norm become local to 2 / 128 adjacent columns (see headnorm_rope_kernel's `paired` form).
enum PackMode { PACK_ROWS = 0, PACK_ROWS_IL32 = 1, PACK_CONV_IN = 2, PACK_CONVT_OUT = 3, PACK_ROWS_HEADPAIR = 4 };
int launch_pack(const void* src, int src_dtype, void* dst, int dst_is_bf16, int mode, long rows, long cols, long dst_ld,
                long dst_row0, int p0, int p1, hipStream_t s);

struct ConvArgs {
    const bf16_t* x; long x_batch_stride; int L_in; int Cin;   // x[b][l][ci] NLC bf16
    // strided (down-sampling) conv as a 2-tap conv over the view x'[r][c'] = x_flat[r*Cin + c' + x_shift], Cin = stride*C:
    // an element is read iff 0 <= flat < x_valid (x_valid == 0: plain conv, rows outside [0, L_in) are zero)
    long x_shift; long x_valid;
    const bf16_t* w;                                             // w[n][tap][ci]
    const float* bias;                                           // [N] or null
    const float* alpha; const float* beta;                       // snake params [Cin] (log scale) or null
    const bf16_t* res; long res_batch_stride;                    // residual, same layout as y, or null
    void* y; long y_batch_stride;                                // out[b][flat]: flat = m*N + n + y_shift (bf16) or NCL f32
    int B, M, N, taps, dil, center;                              // rows m in [0,M): x row = m + (tap-center)*dil
    long y_shift; long y_valid;                                  // flat index valid iff 0 <= flat < y_valid (transposed conv crop)
    int out_mode;                                                // 0: bf16 NLC flat; 1: f32 NCL [b][n][m] (n < n_real)
    int n_real;
    // out_mode 1, windowed (chunked decode): rows m in [ncl_m_lo, ncl_m_hi) land at y[b][n][ncl_off + m - ncl_m_lo] with row pitch
    // ncl_ld; ncl_ld == 0: the whole tensor (pitch M, every row)
    long ncl_ld; long ncl_off; int ncl_m_lo; int ncl_m_hi;
    int wide_ok;                                                 // set by launch_conv: y / res / strides allow the 16-byte staged epilogue
    int clk_probe;                                               // ACE355_CONV_CLK=1: one workgroup records its shader-clock phases
    // fused residual unit (Cin = N = 128, plain conv): y = res + bias2 + w2 . snake2(conv(x) + bias), w2 [N][1][N]: the k = 1
    // conv of an OobleckResidualUnit applied to the k = 7 result while it is still in the workgroup (LDS), launch_conv decides
    const bf16_t* w2; const float* bias2; const float* alpha2; const float* beta2;
    // Snake of the CONSUMER applied by the producer (out_mode 0): y = snake(bias + conv + res) with per-output-channel parameters
    // [N] (exp(alpha), 1 / (exp(beta) + 1e-9)), for activations whose only reader is one Snake -> conv (the k = 7 result of an unfused
    // residual unit, a block's last unit / conv1 ahead of a transposed conv, the last unit ahead of the output conv): the Snake is
    // evaluated once per element from the fp32 sum instead of once per (column tile, halo) of the reader from the bf16-rounded one
    const float* osnake_a; const float* osnake_b;
    // set by launch_conv: XCD-aware rasterisation of the (row block, column tile) grid (ras_tn > 1: a 1-D grid of ras_tm * ras_tn
    // workgroups per batch item, the column tiles of one row block on consecutive slots of ONE XCD); 0: the plain (m, n, b) grid
    int ras_tm; int ras_tn;
};
int launch_conv(const ConvArgs& a, hipStream_t s);
int launch_ncl_to_nlc(const float* z, bf16_t* out, int B, int C, int T, hipStream_t s);

}  // namespace ace355
