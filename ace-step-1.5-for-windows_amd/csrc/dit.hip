// dit.hip - the DiT handle behind ace355.h: weight ingestion/packing, condition slots (cross K/V cache),
// the decoder forward (AceStepDiTModel.forward, base.py:1303-1507) and the sampling loop
// (generate_audio, base.py:1913-1981).  Host-side orchestration only; kernels live in gemm/attn/elementwise.hip.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/ace355.h"
#include "common.h"

using namespace ace355;

namespace {

struct MxW {  // MXFP8 copy of one packed projection (ace355_dit_set_precision): fp8 e4m3 [N, K] + E8M0 block scales [K/128][pad]
    uint8_t* q = nullptr;
    uint32_t* sc = nullptr;
    int pad = 0;
    int bit = 0;   // which projection (ACE355_MX_MASK: 1 qkv, 2 o_proj, 4 gate|up, 8 down, 16 cross q, 32 cross o)
};
struct LayerW {
    bf16_t *wqkv, *wo, *wq_c, *wkv_c, *wo_c, *wgu, *wdown;
    float *n_sa, *n_ca, *n_mlp, *qn_s, *kn_s, *qn_c, *kn_c, *sst;
    MxW mx_qkv, mx_o, mx_gu, mx_down, mx_qc, mx_oc;
};
struct TimeEmbedW {
    bf16_t *l1, *l2, *tp;
    float *b1, *b2, *bp;
};
struct CondSlot {
    int L = 0, Lpad = 0;
    long cap = 0;          // allocated L capacity
    bf16_t* kv = nullptr;  // [layers][L][2*KVD]   (K | V rows as produced by the GEMM; K head-normed in place)
    bf16_t* vt = nullptr;  // [layers][KVH][128][Lpad]
    bool valid = false;
    bool broadcast = false;       // all L keys identical (rows == 1): cross-attention output is a per-layer constant
    float* cross_const = nullptr; // [layers][D] = o_proj(v) of that constant, fp32
};

}  // namespace

struct ace355_dit {
    ace355_dit_config cfg;
    int D, F, QD, KVD, NL, KVH, HQ, OUTC;
    std::vector<LayerW> layers;
    bf16_t *w_in = nullptr, *w_out = nullptr, *w_cond = nullptr;
    float *b_in = nullptr, *b_out = nullptr, *b_cond = nullptr, *norm_out = nullptr, *sst_out = nullptr;
    TimeEmbedW te[2];
    std::set<std::string> loaded;
    size_t expected_tensors = 0;
    bool finalized = false;
    std::vector<void*> allocs;
    void* stage = nullptr;
    size_t stage_bytes = 0;

    // workspace
    int ws_N = 0, ws_T = 0;
    int vt_key_N = -1, vt_key_S = -1;   // (N, S) the pad columns of vt were last zeroed for
    bool zr_on = false; int fwd_M = 0;  // (set by forward_core for its gemm() calls)
    bool dup_half = false;              // set by the sampler around forward_core: sequences [N/2, N) carry the same latents / context / timestep as [0, N/2) (CFG)
    bool dedup_on = true;               // ace355_dit_set_dedup / ACE355_DEDUP0
    int64_t dedup_forwards = 0;         // forwards whose layer 0 ran its QKV projection + self-attention on one half (ace355_dit_dedup_count)
    int zr_M = -1;                      // row zr_M of xn / ao / act is zero: the row the GEMM tiles' pad rows read (GemmEpilogue::a_zero_idx)
    std::vector<void*> ws_allocs;
    bf16_t *xin = nullptr, *xn = nullptr, *qkv = nullptr, *ao = nullptr, *act = nullptr, *vt = nullptr;
    float *h = nullptr, *vpad = nullptr, *tfreq = nullptr, *ta1 = nullptr, *temb = nullptr, *tsilu = nullptr, *tproj = nullptr;
    ModEntry* mod_tab = nullptr;  // [NL][2] (self-attention norm, MLP norm), built at finalize
    float* gs = nullptr;          // [rows][NL][2][2][D] folded (g, sft) vectors of the current forward
    float *xt = nullptr, *avg = nullptr;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    int rope_S = 0;
    // condition staging
    bf16_t *enc_bf = nullptr, *enc_emb = nullptr;
    long enc_cap = 0;
    CondSlot slots[ACE355_MAX_SLOTS];
    int* flags_dev = nullptr;
    float* tap_dst[64] = {};  // ace355_dit_set_tap
    // hipGraph replay of the sampler loop (ace355_dit_set_graph): the captured launch sequence of one whole call, keyed by
    // everything a kernel argument was computed from
    bool graph_mode = false;
    std::string graph_key;
    hipGraphExec_t graph_exec = nullptr;
    hipStream_t graph_stream = nullptr;      // capture / replay stream (the caller's may be the legacy default stream, which
    hipEvent_t graph_in = nullptr, graph_out = nullptr;  // cannot be captured): ordered against the caller's by two events
    long ws_epoch = 0, cond_epoch = 0;   // bumped when workspace / condition-slot memory moves or changes shape
    long graph_replays = 0, graph_captures = 0;
    float *g_ctx_nc = nullptr, *g_sde = nullptr;   // graph mode: handle-owned copies of ctx_non_cover / sde_noise (caller tensors are
    size_t g_ctx_nc_n = 0, g_sde_n = 0;            // re-allocated per call: their addresses must not be part of the graph)
    // MXFP8 mode (BASELINE configs[4] "fp8 MFMA"): the four big projections of every layer run on v_mfma_scale_f32_32x32x64_f8f6f4
    int precision = 0;                 // ACE355_PRECISION_*
    bool weights_fp8wo = false;        // the Linear weights went through the fp8 weight-only round trip (one way: reload to undo)
    std::vector<void*> mx_allocs;      // weight copies
    uint8_t* xq = nullptr;             // activation operand of the current MX GEMM, fp8 [M, max(D, F)]
    uint32_t* xs = nullptr;            // its scales [max(D, F) / 128][xs_pad]
    uint8_t* aq = nullptr;             // SwiGLU output as MXFP8 [M, F] (written by the gate|up GEMM's epilogue, read by the down GEMM)
    uint32_t* as_ = nullptr;           // [F / 128][xs_pad]
    int xs_pad = 0;
    int mx_min_rows = 1536;            // below this many token rows the 192x256 tile does not fill the chip: bf16 kernels

    // RMSNorm folded into the neighbouring GEMM epilogues (sampler path, big-M bf16 launches; GemmEpilogue::nf_* / nc_*): the residual
    // GEMM that finishes h also writes bf16(h * g_next) and the rows' sums of squares, the consumer projection applies rstd and the
    // shift's projection (shift W^T, precomputed for every step of the schedule at the start of the call) to its accumulators.
    struct NormFold {
        int enabled = 1;                 // ACE355_NORM_FOLD (default on)
        int min_rows = 64;               // ACE355_NORM_FOLD_MIN_ROWS: token rows below which the norms stay kernels.  Round 2 kept the small-M
                                         // launches unfolded (1536: at M = 750 folding cost 1.2 % then); with round 3's epilogues it wins
                                         // there too (batch-1 request 149.1 -> 147.9 ms, configs[0] 49.1 -> 48.7 ms in ABAB runs)
        bool on = false;                 // the current sampler call runs folded
        bool emb_on = false;             // the current sampler call reads its per-step timestep embeddings / norm vectors from the tables
        bool bias_ok = false;            // bias tables match `key`
        int step = 0;                    // current step (row of the bias tables)
        int rows = 0;                    // steps of the tables
        int cap_rows = 0; long cap_M = 0;
        std::vector<void*> allocs;
        float *tfreq = nullptr, *ta1 = nullptr, *temb = nullptr, *tsilu = nullptr, *tproj = nullptr, *gs = nullptr;
        bf16_t* shift = nullptr;         // [2 NL][rows][D]
        float* bias_qkv = nullptr;       // [NL][rows][QKV]
        float* bias_gu = nullptr;        // [NL][rows][2F]
        unsigned long long* rowsq = nullptr;  // [NL][3][cap_M] (2^-24 fixed point): self-attention norm, cross-attention norm, MLP norm
        std::vector<float> key;          // schedule the tables were built for ...
        hipStream_t key_stream = nullptr; // ... and the stream that built them (another stream is not ordered behind it)
    } nf;

    // The handle's side stream (chain 2 of the dual-chain sampler below) with its fork / join events and its own split-K turn counters.  (Rounds 4-5 also
    // ran a per-layer "CFG fork" on it - the null rows' MLP beside the conditional rows' cross-attention chain: bit-identical, and slower at every batch
    // size measured (8 songs: 549 vs 507 ms per pass; 1 / 2 songs: 178 vs 141 / 212 vs 190 ms per request, profiles/r06_small_batch_switches.txt) - removed in
    // round 6: tools/r06_cfg_fork.patch, DESIGN.md sections 12.1 / 14.)
    struct CfgFork {
        hipStream_t side = nullptr;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        int* sk_cnt = nullptr;        // the side stream's own split-K turn counters (two concurrent launches must not share tile counters)
    } fk;

    // Dual-chain sampler (round 4).  The songs of a request are independent through the whole sampling loop (generate_audio carries no
    // cross-item term: per-item noise, per-item CFG / APG, base.py:1783-1989), so ace355_dit_sample can run them as TWO half-batch
    // samplers on two hardware queues.  Small requests (2-4 songs of 30 s: launches of a few dozen to ~200 workgroups, latency bound)
    // gain from it - one chain's kernels fill the CUs the other leaves idle; at the metric batch the GEMMs of both chains sit at the
    // power cap and the pair is no faster than one chain (numbers at `max_rows` below and in DESIGN.md section 10).  Chain 2 runs on a
    // CONTEXT: a second ace355_dit that aliases this handle's weights and condition slots and owns its workspace and schedule tables.
    struct Dual {
        int mode = 1;                 // ACE355_DUAL / ace355_dit_set_dual: 0 one chain; 1 (default) two chains for requests of >= 2 songs whose
                                      // one-chain launches would under-fill the chip (<= max_rows token rows in all, or a tile count that leaves
                                      // > 15 % of the CU slots of its rounds empty: ace355_dit_sample); 2 two chains whenever >= 2 songs
        int max_rows = 2400;          // ACE355_DUAL_MAX_ROWS (token rows of the whole request, CFG copies included).  Measured (same-box ABAB x 2-3,
                                      // 30 s songs, DiT + decode, ms per request, one chain -> two): 2 songs (1500 rows) 201.2 -> 197.5 (and 207.1 ->
                                      // 199.1, 209.8 -> 201.0 on other boxes), 3 songs (2250 rows: 12 row tiles, an awkward fill) 275.9 -> 253.3,
                                      // 4 songs (3000 rows) 300.1 -> 315.4, 8 songs 507.7 -> 513-528: where one chain's launches fill the chip
                                      // the GEMMs sit at the power cap and two chains only share it
        int slots_min_rows = 1 << 30; // ACE355_DUAL_SLOTS_MIN_ROWS: chains with at least this many token rows plan their launches for half the
                                      // chip (cu_slots 128).  Off by default: 473 ms against 479 without it for two 4-song chains (DiT only),
                                      // but slower for small chains (2 songs: 226 vs 199 ms)
        ace355_dit* ctx = nullptr;    // chain 2's context (created on first use)
        std::vector<std::pair<hipStream_t, bool>> probed;   // caller streams the side stream was checked against -> on a queue of its own?
        bool concurrent = false;      // the answer for the current call's stream
        long calls = 0;               // sampler calls that ran as two chains (tests)
    } dual;
    bool pf_block = false;   // set by ace355_dit_sample around a two-chain call: the other chain's launches want the CUs a prefetch would take
    struct { const void* w = nullptr; int M = 0, N = 0, K = 0, mode = 0; } pf_next;   // set by forward_core before a gemm(): the NEXT projection (weight prefetch)
    bool alias = false;      // this object is a chain context: weights, slots and rope tables belong to the owning handle
    int cu_slots = 0;        // GemmEpilogue::cu_slots / AttnArgs::cu_slots of this context's launches (0: the whole chip)

    int* sk_cnt = nullptr;   // split-K counters lent to launch_gemm (GemmEpilogue::sk_cnt)
    float* attn_part = nullptr;   // split-KV scratch lent to launch_attention (AttnArgs::part): small problems only
    long attn_part_floats = 0;

    // profiling
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gemm_ev, attn_ev;
    double gemm_flops = 0, attn_flops = 0;
    long gemm_launches = 0;
};

namespace {

template <typename T>
int dev_alloc(std::vector<void*>& bag, T** p, size_t n) {
    void* q = nullptr;
    ACE_HIP(hipMalloc(&q, n * sizeof(T) + 256));
    bag.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}
#define ALLOC(bag, ptr, n)                         \
    do {                                           \
        int _rc = dev_alloc(bag, &(ptr), (n));     \
        if (_rc) return _rc;                       \
    } while (0)

struct Dest {
    void* dst;
    int is_bf16;
    int mode;
    long rows, cols, dst_ld, dst_row0;
    int p0, p1;
};

// Map a reference state_dict key to its packed destination.
bool resolve(ace355_dit* h, const std::string& name, Dest* d) {
    const long D = h->D, F = h->F, QD = h->QD, KVD = h->KVD;
    auto rows = [&](void* dst, int bf, long r, long c, long ld, long row0) {
        *d = Dest{dst, bf, PACK_ROWS, r, c, ld, row0, 0, 0};
        return true;
    };
    if (name.rfind("layers.", 0) == 0) {
        const size_t dot = name.find('.', 7);
        if (dot == std::string::npos) return false;
        const int li = atoi(name.substr(7, dot - 7).c_str());
        if (li < 0 || li >= h->NL) return false;
        LayerW& L = h->layers[li];
        const std::string r = name.substr(dot + 1);
        if (r == "scale_shift_table") return rows(L.sst, 0, 6, D, D, 0);
        if (r == "self_attn_norm.weight") return rows(L.n_sa, 0, 1, D, D, 0);
        if (r == "cross_attn_norm.weight") return rows(L.n_ca, 0, 1, D, D, 0);
        if (r == "mlp_norm.weight") return rows(L.n_mlp, 0, 1, D, D, 0);
        // self-attention q / k rows in the head-pair order (common.h PackMode): RoPE partners adjacent
        if (r == "self_attn.q_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, QD, D, D, 0, 0, 0}; return true; }
        if (r == "self_attn.k_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, KVD, D, D, QD, 0, 0}; return true; }
        if (r == "self_attn.v_proj.weight") return rows(L.wqkv, 1, KVD, D, D, QD + KVD);
        if (r == "self_attn.o_proj.weight") return rows(L.wo, 1, D, QD, QD, 0);
        if (r == "self_attn.q_norm.weight") return rows(L.qn_s, 0, 1, 128, 128, 0);
        if (r == "self_attn.k_norm.weight") return rows(L.kn_s, 0, 1, 128, 128, 0);
        if (r == "cross_attn.q_proj.weight") return rows(L.wq_c, 1, QD, D, D, 0);
        if (r == "cross_attn.k_proj.weight") return rows(L.wkv_c, 1, KVD, D, D, 0);
        if (r == "cross_attn.v_proj.weight") return rows(L.wkv_c, 1, KVD, D, D, KVD);
        if (r == "cross_attn.o_proj.weight") return rows(L.wo_c, 1, D, QD, QD, 0);
        if (r == "cross_attn.q_norm.weight") return rows(L.qn_c, 0, 1, 128, 128, 0);
        if (r == "cross_attn.k_norm.weight") return rows(L.kn_c, 0, 1, 128, 128, 0);
        if (r == "mlp.gate_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 0, 0}; return true; }
        if (r == "mlp.up_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 1, 0}; return true; }
        if (r == "mlp.down_proj.weight") return rows(L.wdown, 1, D, F, F, 0);
        return false;
    }
    const int C = h->cfg.in_channels, P = h->cfg.patch_size, OC = h->OUTC;
    if (name == "proj_in.1.weight") { *d = Dest{h->w_in, 1, PACK_CONV_IN, D, (long)C * P, (long)C * P, 0, C, P}; return true; }
    if (name == "proj_in.1.bias") return rows(h->b_in, 0, 1, D, D, 0);
    if (name == "proj_out.1.weight") { *d = Dest{h->w_out, 1, PACK_CONVT_OUT, D, (long)OC * P, D, 0, OC, P}; return true; }
    if (name == "proj_out.1.bias") return rows(h->b_out, 0, 1, OC, OC, 0);  // duplicated per patch position at finalize
    for (int e = 0; e < 2; ++e) {
        const std::string p = e == 0 ? "time_embed." : "time_embed_r.";
        if (name.rfind(p, 0) != 0) continue;
        const std::string r = name.substr(p.size());
        TimeEmbedW& T = h->te[e];
        if (r == "linear_1.weight") return rows(T.l1, 1, D, 256, 256, 0);
        if (r == "linear_1.bias") return rows(T.b1, 0, 1, D, D, 0);
        if (r == "linear_2.weight") return rows(T.l2, 1, D, D, D, 0);
        if (r == "linear_2.bias") return rows(T.b2, 0, 1, D, D, 0);
        if (r == "time_proj.weight") return rows(T.tp, 1, 6 * D, D, D, 0);
        if (r == "time_proj.bias") return rows(T.bp, 0, 1, 6 * D, 6 * D, 0);
        return false;
    }
    if (name == "condition_embedder.weight") return rows(h->w_cond, 1, D, D, D, 0);
    if (name == "condition_embedder.bias") return rows(h->b_cond, 0, 1, D, D, 0);
    if (name == "norm_out.weight") return rows(h->norm_out, 0, 1, D, D, 0);
    if (name == "scale_shift_table") return rows(h->sst_out, 0, 2, D, D, 0);
    return false;
}

struct EvScope {
    ace355_dit* h;
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* bag;
    hipStream_t s;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    EvScope(ace355_dit* h_, std::vector<std::pair<hipEvent_t, hipEvent_t>>* b, hipStream_t s_) : h(h_), bag(b), s(s_) {
        if (h->profile) {
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, s);
        }
    }
    ~EvScope() {
        if (h->profile) {
            hipEventRecord(e1, s);
            bag->push_back({e0, e1});
        }
    }
};

int gemm(ace355_dit* h, const bf16_t* A, int lda, const bf16_t* W, int ldw, void* C, int ldc, int M, int N, int K,
         const GemmEpilogue& ep, hipStream_t s) {
    EvScope ev(h, &h->gemm_ev, s);
    if (h->profile) {
        h->gemm_flops += 2.0 * M * N * K;
        h->gemm_launches++;
    }
    GemmEpilogue e2 = ep;
    const bool side = h->fk.side && s == h->fk.side;
    e2.sk_cnt = side ? h->fk.sk_cnt : h->sk_cnt;
    e2.cu_slots = h->cu_slots;
    if (h->pf_next.w) {   // the GEMM that follows this one in forward_core's sequence: its weight rows start towards the L2s now (GemmEpilogue::pf_*)
        gemm_prefetch_plan(&e2, h->pf_next.w, h->pf_next.M, h->pf_next.N, h->pf_next.K, h->pf_next.mode, h->cu_slots);
        h->pf_next.w = nullptr;
    }
    // pad rows of the last row tile read the workspace's zero row (forward_core keeps row fwd_M of xn / ao / act zero)
    if (h->zr_on && h->zr_M == h->fwd_M && h->fwd_M >= M) {
        const struct { const bf16_t* base; int ld; } ops[3] = {{h->xn, h->D}, {h->ao, h->QD}, {h->act, h->F}};
        for (const auto& o : ops) {
            if (lda != o.ld || A < o.base || A >= o.base + (size_t)h->fwd_M * o.ld) continue;
            const size_t off = (size_t)(A - o.base);
            if (off % o.ld == 0 && (long)(off / o.ld) + M <= h->fwd_M) e2.a_zero_idx = h->fwd_M - (int)(off / o.ld);
        }
    }
    return launch_gemm(A, lda, W, ldw, C, ldc, M, N, K, e2, s);
}

// The same projection on the MXFP8 path: the bf16 activation operand is block-quantised (mx_quant_kernel) into the handle's scratch,
// then the MX GEMM runs against the layer's fp8 weight copy with the bf16 kernel's epilogue.  Caller checked mx_usable().
// A == nullptr: the producer already left the operand in the scratch as MXFP8 (launch_rmsnorm_gs_mx).
int gemm_mx(ace355_dit* h, const bf16_t* A, int lda, const MxW& W, void* C, int ldc, int M, int N, int K, const GemmEpilogue& ep,
            hipStream_t s, const uint8_t* preq = nullptr, const uint32_t* pres = nullptr) {
    if (A) {
        int rc = launch_mx_quant(A, lda, M, K, h->xq, h->xs, h->xs_pad, s);
        if (rc) return rc;
    }
    GemmEpilogue epx = ep;
    epx.cu_slots = h->cu_slots;
    if (preq) {
        EvScope ev(h, &h->gemm_ev, s);
        if (h->profile) {
            h->gemm_flops += 2.0 * M * N * K;
            h->gemm_launches++;
        }
        return launch_gemm_mx(preq, pres, h->xs_pad, W.q, W.sc, W.pad, C, ldc, M, N, K, epx, s);
    }
    EvScope ev(h, &h->gemm_ev, s);
    if (h->profile) {
        h->gemm_flops += 2.0 * M * N * K;
        h->gemm_launches++;
    }
    return launch_gemm_mx(h->xq, h->xs, h->xs_pad, W.q, W.sc, W.pad, C, ldc, M, N, K, epx, s);
}
bool mx_usable(const ace355_dit* h, const MxW& W, int M, int N, int K, int mode, int q_cols = 0, int qk_cols = 0) {
    static int mask = -1;  // ACE355_MX_MASK: subset of the projections that run in MXFP8 (error / time trade-offs)
    if (mask < 0) { const char* e = getenv("ACE355_MX_MASK"); mask = e ? atoi(e) : 63; }
    return h->precision == ACE355_PRECISION_MXFP8 && W.q && (mask & W.bit) && M >= h->mx_min_rows && gemm_mx_supported(M, N, K, mode) &&
           (mode != 4 || (q_cols % 256 == 0 && qk_cols % 256 == 0));
}

int ensure_rope(ace355_dit* h, int S, hipStream_t s) {
    if (S <= h->rope_S) return 0;
    int cap = 512;
    while (cap < S) cap *= 2;
    if (h->rope_cos) {  // growing: the old tables may still be read by queued work
        ACE_HIP(hipStreamSynchronize(s));
        hipFree(h->rope_cos);
        hipFree(h->rope_sin);
        h->rope_cos = h->rope_sin = nullptr;
        h->rope_S = 0;
    }
    ACE_HIP(hipMalloc((void**)&h->rope_cos, (size_t)cap * 64 * sizeof(float)));
    ACE_HIP(hipMalloc((void**)&h->rope_sin, (size_t)cap * 64 * sizeof(float)));
    int rc = launch_rope_table(h->rope_cos, h->rope_sin, cap, h->cfg.rope_theta, s);
    if (rc) return rc;
    h->ws_epoch++;
    h->rope_S = cap;
    return 0;
}

int ensure_workspace(ace355_dit* h, int N, int T, hipStream_t s) {
    const int S = (T + 1) / 2;
    int rc = ensure_rope(h, S, s);
    if (rc) return rc;
    if (N <= h->ws_N && T <= h->ws_T) return 0;
    ACE_HIP(hipStreamSynchronize(s));
    for (void* p : h->ws_allocs) hipFree(p);
    h->ws_allocs.clear();
    const int capN = N > h->ws_N ? N : h->ws_N, capT = T > h->ws_T ? T : h->ws_T;
    const long cS = (capT + 1) / 2, cTp = 2 * cS, M = (long)capN * cS, Sp = ((cS + 63) / 64) * 64;
    const long D = h->D, QKV = h->QD + 2 * h->KVD;
    ALLOC(h->ws_allocs, h->xin, (size_t)capN * cTp * 192);
    ALLOC(h->ws_allocs, h->h, (size_t)M * D);
    ALLOC(h->ws_allocs, h->xn, (size_t)(M + 1) * D);      // (+ 1: the zero row behind the rows of the call in progress, forward_core)
    ALLOC(h->ws_allocs, h->qkv, (size_t)M * QKV);
    ALLOC(h->ws_allocs, h->ao, (size_t)(M + 1) * h->QD);
    ALLOC(h->ws_allocs, h->act, (size_t)(M + 1) * h->F);
    ALLOC(h->ws_allocs, h->vt, (size_t)capN * h->KVH * 128 * Sp);
    ALLOC(h->ws_allocs, h->vpad, (size_t)M * 2 * h->OUTC);
    ALLOC(h->ws_allocs, h->tfreq, (size_t)2 * capN * 256);
    ALLOC(h->ws_allocs, h->ta1, (size_t)capN * D);
    ALLOC(h->ws_allocs, h->temb, (size_t)capN * D);
    ALLOC(h->ws_allocs, h->tsilu, (size_t)capN * D);
    ALLOC(h->ws_allocs, h->tproj, (size_t)capN * 6 * D);
    ALLOC(h->ws_allocs, h->gs, (size_t)capN * h->NL * 4 * D);
    ALLOC(h->ws_allocs, h->xt, (size_t)capN * capT * h->OUTC);
    ALLOC(h->ws_allocs, h->avg, (size_t)capN * capT * h->OUTC);
    {
        const long KX = D > h->F ? D : h->F;
        h->xs_pad = mx_rows_pad((int)M);
        ALLOC(h->ws_allocs, h->xq, (size_t)M * KX + 256);
        ALLOC(h->ws_allocs, h->xs, (size_t)(KX / 128 + 1) * h->xs_pad);
        ACE_HIP(hipMemsetAsync(h->xs, 0, (size_t)(KX / 128 + 1) * h->xs_pad * sizeof(uint32_t), s));
        ALLOC(h->ws_allocs, h->aq, (size_t)M * h->F + 256);
        ALLOC(h->ws_allocs, h->as_, (size_t)(h->F / 128 + 1) * h->xs_pad);
        ACE_HIP(hipMemsetAsync(h->as_, 0, (size_t)(h->F / 128 + 1) * h->xs_pad * sizeof(uint32_t), s));
    }
    h->ws_N = capN;
    h->ws_T = capT;
    h->vt_key_N = h->vt_key_S = -1;
    h->zr_M = -1;
    h->ws_epoch++;
    return 0;
}

// TimestepEmbedding x2 (base.py:1340-1344): temb[rows][D], tproj[rows][6D] for `rows` distinct (t, t - t_r) pairs.
int time_embed_into(ace355_dit* h, const float* t, const float* tr, int rows, float* tfreq, float* ta1, float* temb, float* tsilu,
                    float* tproj, hipStream_t s) {
    TVals tv;
    const int D = h->D;
    ACE_CHECK(rows <= 64, "time_embed: at most 64 rows per call");
    for (int e = 0; e < 2; ++e) {
        for (int i = 0; i < rows; ++i) tv.t[i] = e == 0 ? t[i] : (t[i] - tr[i]);
        float* tf = tfreq + (size_t)e * rows * 256;
        int rc = launch_sinusoid(tv, rows, tf, s);
        if (rc) return rc;
        const TimeEmbedW& T = h->te[e];
        rc = launch_small_linear_ex(tf, T.l1, T.b1, ta1, nullptr, rows, D, 256, /*silu_out*/ 1, 0, s);
        if (rc) return rc;
        rc = launch_small_linear_ex(ta1, T.l2, T.b2, temb, tsilu, rows, D, D, 0, /*accumulate*/ e, s);
        if (rc) return rc;
        rc = launch_small_linear_ex(tsilu, T.tp, T.bp, tproj, nullptr, rows, 6 * D, D, 0, e, s);
        if (rc) return rc;
    }
    return 0;
}
int time_embed(ace355_dit* h, const float* t, const float* tr, int rows, hipStream_t s) {
    return time_embed_into(h, t, tr, rows, h->tfreq, h->ta1, h->temb, h->tsilu, h->tproj, s);
}

// Folded RMSNorm, per sampler call: decide whether this call runs folded and (re)build the per-step bias tables
// bias[li][i][:] = shift_i W^T for the QKV and gate|up projections (one M = steps GEMM per layer and projection: one pass over those
// weights per CALL; as per-step GEMVs it would be one pass per forward, which costs what the norm kernels cost).
bool normfold_eligible(const ace355_dit* h, int steps, int M) {
    const int D = h->D, F = h->F, QD = h->QD, QKV = h->QD + 2 * h->KVD;
    const bool mx = h->precision == ACE355_PRECISION_MXFP8 && M >= h->mx_min_rows;
    return h->nf.enabled && gemm_fold_supported() && !mx && M >= (h->nf.enabled >= 2 ? 64 : h->nf.min_rows) && D % 256 == 0 && QKV % 256 == 0 && QD % 256 == 0 && (2 * F) % 256 == 0 && h->KVD % 128 == 0 &&
           steps <= 1024;
}
// (allocation: outside any stream capture, next to ensure_workspace)
int normfold_reserve(ace355_dit* h, int steps, int N, int T, hipStream_t s) {
    auto& nf = h->nf;
    const int S = (T + 1) / 2, M = N * S;
    if (steps > 1024 || (steps <= nf.cap_rows && M <= nf.cap_M)) return 0;   // (every sampler call uses the embedding tables; the fold its bias tables)
    const int D = h->D, F = h->F, QKV = h->QD + 2 * h->KVD, NL = h->NL;
    ACE_HIP(hipStreamSynchronize(s));
    const size_t R = std::max(steps, nf.cap_rows), MM = std::max<long>(M, nf.cap_M);
    for (void* q : nf.allocs) hipFree(q);
    nf.allocs.clear();
    nf.key.clear();
    nf.cap_rows = 0; nf.cap_M = 0;   // (an allocation failure below must not leave capacities that point at freed buffers)
    ALLOC(nf.allocs, nf.tfreq, 2 * R * 256);
    ALLOC(nf.allocs, nf.ta1, R * D);
    ALLOC(nf.allocs, nf.temb, R * D);
    ALLOC(nf.allocs, nf.tsilu, R * D);
    ALLOC(nf.allocs, nf.tproj, R * 6 * D);
    ALLOC(nf.allocs, nf.gs, R * NL * 4 * D);
    ALLOC(nf.allocs, nf.shift, (size_t)2 * NL * R * D);
    ALLOC(nf.allocs, nf.bias_qkv, (size_t)NL * R * QKV);
    ALLOC(nf.allocs, nf.bias_gu, (size_t)NL * R * 2 * F);
    ALLOC(nf.allocs, nf.rowsq, (size_t)NL * 3 * MM);
    nf.cap_rows = (int)R;
    nf.cap_M = (long)MM;
    h->ws_epoch++;   // a captured graph holds these addresses
    return 0;
}
int normfold_prepare(ace355_dit* h, const ace355_sample_params* p, int N, int T, hipStream_t s) {
    auto& nf = h->nf;
    nf.on = false;
    nf.emb_on = false;
    const int S = (T + 1) / 2, M = N * S, steps = p->num_steps;
    const int D = h->D, F = h->F, QKV = h->QD + 2 * h->KVD, NL = h->NL;
    if (steps > nf.cap_rows || M > nf.cap_M) return 0;
    static int tab_env = -1;   // ACE355_SCHED_TABLES=0: per-step time_embed / mod_gs launches as in round 1 (A/B; also switches the fold off)
    if (tab_env < 0) { const char* e = getenv("ACE355_SCHED_TABLES"); tab_env = e ? atoi(e) : 1; }
    if (!tab_env) return 0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap != hipStreamCaptureStatusNone) nf.key.clear();   // a captured call always carries its own table build
    std::vector<float> key(p->t_sched_host, p->t_sched_host + steps);
    int rc;
    if (key != nf.key || nf.rows != steps || nf.key_stream != s) {
        // TimestepEmbedding x2 and the folded norm vectors of the WHOLE schedule (64 rows per launch group): the per-step launches
        // of the loop (8 small linears + mod_gs) become reads of row i
        for (int r0 = 0; r0 < steps; r0 += 64) {
            const int nr = std::min(64, steps - r0);
            // (the two embeddings' sinusoid rows of a chunk sit side by side in tfreq: every chunk gets its own 2 x 64 x 256 block)
            rc = time_embed_into(h, p->t_sched_host + r0, p->t_sched_host + r0, nr, nf.tfreq + (size_t)2 * r0 * 256, nf.ta1 + (size_t)r0 * D,
                                 nf.temb + (size_t)r0 * D, nf.tsilu + (size_t)r0 * D, nf.tproj + (size_t)r0 * 6 * D, s);
            if (rc) return rc;
        }
        rc = launch_mod_gs(h->mod_tab, 2 * NL, nf.tproj, 6L * D, steps, nf.gs, D, s);
        if (rc) return rc;
        nf.key = key;
        nf.rows = steps;
        nf.key_stream = s;
        nf.bias_ok = false;
    }
    nf.emb_on = true;
    if (!normfold_eligible(h, steps, M)) return 0;
    if (!nf.bias_ok) {
        rc = launch_shift_rows(nf.gs, 2 * NL, steps, nf.shift, D, s);
        if (rc) return rc;
        for (int li = 0; li < NL; ++li) {
            const LayerW& W = h->layers[li];
            GemmEpilogue ep{1, nullptr, nullptr, nullptr, 0, 0};
            rc = gemm(h, nf.shift + (size_t)(li * 2 + 0) * steps * D, D, W.wqkv, D, nf.bias_qkv + (size_t)li * steps * QKV, QKV, steps, QKV, D, ep, s);
            if (rc) return rc;
            rc = gemm(h, nf.shift + (size_t)(li * 2 + 1) * steps * D, D, W.wgu, D, nf.bias_gu + (size_t)li * steps * 2 * F, 2 * F, steps, 2 * F, D, ep, s);
            if (rc) return rc;
        }
        nf.bias_ok = true;
    }
    nf.on = true;
    nf.step = 0;
    return 0;
}

// Decoder body on a packed patch input xin[N][Tpad][192]; writes vpad[N][Tpad][64].
// temb_rows == 1: every sequence shares one timestep (the sampler); else one row per sequence.
int forward_core(ace355_dit* h, int N, int T, const int* slots, int temb_rows, hipStream_t s) {
    const int S = (T + 1) / 2, M = N * S;
    const int D = h->D, F = h->F, QD = h->QD, KVD = h->KVD, QKV = QD + 2 * KVD;
    const int Sp = ((S + 63) / 64) * 64;
    const float eps = h->cfg.rms_norm_eps;
    const float scale = 1.0f / sqrtf((float)h->cfg.head_dim);
    const int tstride = temb_rows == 1 ? 0 : 6 * D;
    int rc;
    GemmEpilogue ep{};

    for (int i = 0; i < N; ++i)
        ACE_CHECK(slots[i] >= 0 && slots[i] < ACE355_MAX_SLOTS && h->slots[slots[i]].valid, "forward: condition slot not set");
    // trailing sequences that attend a broadcast slot (the CFG null branch) skip cross-attention entirely: their
    // residual term is the slot's per-layer constant, folded into the self-attention epilogue below.
    int n_sc = 0;
    while (n_sc < N && h->slots[slots[N - 1 - n_sc]].broadcast) ++n_sc;
    for (int i = N - n_sc; i < N; ++i)
        if (slots[i] != slots[N - 1]) { n_sc = 0; break; }  // one constant per call
    const int Nc = N - n_sc, Mc = Nc * S;
    // the sequences that DO run cross-attention share one key count; the broadcast slot of the shortcut has no say (L identical keys give
    // the same constant for every L: the null slot built for the cover conditions' length serves the non-cover phase of another length,
    // as null_condition_emb.expand_as(...) does for either, base.py:1907 / :1919)
    int L = -1;
    for (int i = 0; i < (Nc > 0 ? Nc : N); ++i) {
        if (L < 0) L = h->slots[slots[i]].L;
        ACE_CHECK(h->slots[slots[i]].L == L, "forward: the condition slots that run cross-attention in one call must share L");
    }
    const int Lpad = ((L + 63) / 64) * 64;
    const float* cconst = n_sc ? h->slots[slots[N - 1]].cross_const : nullptr;

    if (h->vt_key_N != N || h->vt_key_S != S) {   // V^T layout [N][KVH][128][Sp] changed: pad positions [S, Sp) must read as zero
        ACE_HIP(hipMemsetAsync(h->vt, 0, (size_t)N * h->KVH * 128 * Sp * sizeof(bf16_t), s));
        h->vt_key_N = N; h->vt_key_S = S;
    }
    // Zero row behind the M rows of the three bf16 GEMM operands (GemmEpilogue::a_zero_idx): written here whenever M changes - a larger
    // call since then used that row as data - and by nothing else (every producer stores rows < M only).
    static int zr_env = -1;
    if (zr_env < 0) { const char* e = getenv("ACE355_GEMM_ZROW"); zr_env = e ? atoi(e) : 1; }
    // (under graph capture ALWAYS: a graph recorded while zr_M == M would hold no memset, and replaying it after a larger call used row M as
    //  data would read stale pad rows - harmless for the results, the rows are discarded, but not the zeros the comment above promises; advisor r5)
    if (zr_env && (h->zr_M != M || h->graph_mode)) {
        ACE_HIP(hipMemsetAsync(h->xn + (size_t)M * D, 0, (size_t)D * sizeof(bf16_t), s));
        ACE_HIP(hipMemsetAsync(h->ao + (size_t)M * QD, 0, (size_t)QD * sizeof(bf16_t), s));
        ACE_HIP(hipMemsetAsync(h->act + (size_t)M * F, 0, (size_t)F * sizeof(bf16_t), s));
        h->zr_M = M;
    }
    h->zr_on = zr_env != 0;
    h->fwd_M = M;
    const long gs_stride = temb_rows == 1 ? 0 : (long)h->NL * 4 * D;
    // folded RMSNorm (sampler path): row sums of squares of the 3 NL norm inputs accumulate during this forward
    const bool fold = h->nf.on && temb_rows == 1 && M <= h->nf.cap_M;
    const auto& nf = h->nf;
    auto rowsq = [&](int li, int which) { return nf.rowsq + ((size_t)li * 3 + which) * M; };   // (M <= cap_M: this forward's rows)
    if (fold) ACE_HIP(hipMemsetAsync(nf.rowsq, 0, (size_t)h->NL * 3 * M * sizeof(unsigned long long), s));
    const float inv_d = 1.0f / (float)D;
    // (a sampler call computed the TimestepEmbedding and the folded norm vectors of EVERY step up front: this step's rows are used in
    //  place, and the loop has no per-step time_embed / mod_gs launches)
    const bool emb = nf.emb_on && temb_rows == 1;
    const float* tproj_p = emb ? nf.tproj + (size_t)nf.step * 6 * D : h->tproj;
    const float* temb_p = emb ? nf.temb + (size_t)nf.step * D : h->temb;
    const float* gs_p = emb ? nf.gs + (size_t)nf.step * h->NL * 4 * D : h->gs;
    if (!emb) {
        // fold w * (1 + scale) and shift of the 2 * NL modulated norms for this forward's timestep rows (one launch)
        rc = launch_mod_gs(h->mod_tab, 2 * h->NL, h->tproj, 6L * D, temb_rows, h->gs, D, s);
        if (rc) return rc;
    }

    // patchify: Conv1d(192 -> D, k=2, s=2) == GEMM over [M, 384] (base.py:1358)
    ep = GemmEpilogue{1, h->b_in, nullptr, nullptr, 0, 0};
    rc = gemm(h, h->xin, 2 * h->cfg.in_channels, h->w_in, 2 * h->cfg.in_channels, h->h, D, M, D, 2 * h->cfg.in_channels, ep, s);
    if (rc) return rc;

    // Weight prefetch of the one-song launches (gemm_prefetch_plan, gemm.hip): every projection tells its launch which projection follows
    const bool pf_on = h->precision != ACE355_PRECISION_MXFP8 && !h->pf_block;
    RoctxRange r_fwd("ace355.dit_forward");
    for (int li = 0; li < h->NL; ++li) {
        RoctxRange r_layer("ace355.dit_layer");
        const LayerW& W = h->layers[li];
        const bool sliding = (h->cfg.sliding_layer_mask >> li) & 1ull;
        // the projection that FOLLOWS the next gemm() call: its weight rows are prefetched by that launch's spare workgroups (gemm(): pf_next)
        auto next_gemm = [&](const bf16_t* w, int m, int n, int k, int mode) {
            if (pf_on && m > 0) { h->pf_next.w = w; h->pf_next.M = m; h->pf_next.N = n; h->pf_next.K = k; h->pf_next.mode = mode; }
        };
        // ---- self attention (base.py:499-511)
        // Layer 0 of a CFG forward: the conditional and the null sequence of a song enter the decoder with the same latents, context and
        // timestep (x = cat([xt, xt]), base.py:1929) and differ only from the first cross-attention on - the reference computes the same
        // numbers twice.  Here the first norm, the QKV projection and the self-attention of layer 0 run on the conditional half only, and
        // the o_proj GEMM reads that half's attention output for both (GemmEpilogue::a_wrap); its epilogue (gate, constant term of the
        // null rows, the two folded norms) is per row as always.  ace355_dit_set_dedup(h, 0) / ACE355_DEDUP0=0 switch it off (A/B, tests).
        const bool dedup0 = li == 0 && h->dup_half && h->dedup_on && temb_rows == 1 && Nc > 0 && N == 2 * Nc && h->precision != ACE355_PRECISION_MXFP8 &&
                            !h->tap_dst[0] && gemm_fold_supported();
        const int Mq = dedup0 ? Mc : M, Nq = dedup0 ? Nc : N;   // rows / sequences the self-attention half of this layer computes
        if (dedup0) h->dedup_forwards++;
        const bool mx_qkv = mx_usable(h, W.mx_qkv, M, QKV, D, 4, QD, QD + KVD) && D == 2048;
        const bool fold_sa = fold && li > 0;   // layer 0's input comes from the patchify GEMM: its norm stays a kernel
        if (fold_sa) rc = 0;                   // xn = bf16(h * g) and the row sums were written by the previous layer's down projection
        else if (mx_qkv)  // the norm writes the MX GEMM's operand directly (fp8 + block scales): no bf16 xn round trip, no quantise pass
            rc = launch_rmsnorm_gs_mx(h->h, gs_p + (size_t)(li * 2 + 0) * 2 * D, gs_p + (size_t)(li * 2 + 0) * 2 * D + D, h->xq, h->xs, h->xs_pad,
                                      M, D, eps, gs_stride, S, s);
        else
        rc = launch_rmsnorm_gs(h->h, gs_p + (size_t)(li * 2 + 0) * 2 * D, gs_p + (size_t)(li * 2 + 0) * 2 * D + D, h->xn, Mq, D, eps,
                               gs_stride, S, s);
        if (rc) return rc;
        // QKV projection with q / k head-norm + RoPE in its epilogue (mode 4; launch_gemm falls back to two kernels)
        ep = GemmEpilogue{4, nullptr, nullptr, nullptr, 0, S};
        ep.hn_wq = W.qn_s, ep.hn_wk = W.kn_s, ep.hn_cos = h->rope_cos, ep.hn_sin = h->rope_sin;
        ep.hn_q_cols = QD, ep.hn_qk_cols = QD + KVD, ep.hn_eps = eps;
        if (fold_sa) {
            ep.nc_rowsq = rowsq(li, 0); ep.nc_bias = nf.bias_qkv + ((size_t)li * nf.rows + nf.step) * QKV;
            ep.nc_inv_d = inv_d; ep.nc_eps = eps;
        }
        int vt_done = 0;   // small-M launches write V^T from the QKV epilogue (GemmEpilogue::vt_out): no transpose_v launch
        ep.vt_out = h->vt; ep.vt_ld = Sp; ep.vt_heads = h->KVH; ep.vt_done = &vt_done;
        next_gemm(W.wo, M, D, QD, 2);
        if (mx_qkv) rc = gemm_mx(h, nullptr, D, W.mx_qkv, h->qkv, QKV, M, QKV, D, ep, s);
        else if (mx_usable(h, W.mx_qkv, M, QKV, D, 4, QD, QD + KVD)) rc = gemm_mx(h, h->xn, D, W.mx_qkv, h->qkv, QKV, M, QKV, D, ep, s);
        else rc = gemm(h, h->xn, D, W.wqkv, D, h->qkv, QKV, Mq, QKV, D, ep, s);
        if (rc) return rc;
        if (!vt_done) {
            rc = launch_transpose_v(h->qkv, QKV, QD + KVD, Nq, S, h->KVH, h->vt, Sp, s);
            if (rc) return rc;
        }
        bool ao_is_mx = false;
        {
            AttnArgs a{};
            a.q = h->qkv; a.q_seq_stride = (long)S * QKV; a.q_row_stride = QKV;
            a.k = h->qkv + QD; a.k_seq_stride = (long)S * QKV; a.k_head_stride = 128; a.k_row_stride = QKV;
            a.vt = h->vt; a.vt_seq_stride = (long)h->KVH * 128 * Sp; a.vt_head_stride = 128L * Sp; a.vt_ld = Sp;
            a.use_tab = 0;
            a.out = h->ao; a.o_seq_stride = (long)S * QD; a.o_row_stride = QD;
            a.N = Nq; a.Sq = S; a.Skv = S; a.Hq = h->HQ; a.Hkv = h->KVH;
            a.window = sliding ? h->cfg.sliding_window : -1;
            a.scale = scale;
            a.part = h->attn_part; a.part_floats = h->attn_part_floats;
            a.cu_slots = h->cu_slots;
            if (mx_usable(h, W.mx_o, M, D, QD, 2) && attention_mx_out_ok(a)) {  // the o_proj MX GEMM's operand straight from the attention epilogue
                a.out_q = h->xq; a.out_scales = h->xs; a.out_pad = h->xs_pad;
                ao_is_mx = true;
            }
            EvScope ev(h, &h->attn_ev, s);
            if (h->profile) {
                double keys = (double)S;
                if (sliding) {
                    double tot = 0;
                    for (int i = 0; i < S; ++i) tot += (double)(std::min(S - 1, i + a.window) - std::max(0, i - a.window) + 1);
                    keys = tot / S;
                }
                h->attn_flops += 4.0 * Nq * h->HQ * (double)S * keys * 128.0;
            }
            rc = launch_attention(a, s);
            if (rc) return rc;
        }
        ep = GemmEpilogue{2, nullptr, W.sst + 2 * D, tproj_p + 2 * D, tstride, S, cconst ? cconst + (size_t)li * D : nullptr, Mc};
        const float* g_mlp = gs_p + (size_t)(li * 2 + 1) * 2 * D;
        if (fold) {  // conditional rows go on to the cross-attention norm (plain weight), the others straight to the MLP norm
            ep.nf_xg = h->xn; ep.nf_ldx = D; ep.nf_split = Nc > 0 ? Mc : 0;
            ep.nf_gA = W.n_ca; ep.nf_sqA = rowsq(li, 1);
            ep.nf_gB = g_mlp; ep.nf_sqB = rowsq(li, 2);
        }
        if (dedup0) ep.a_wrap = Mc;   // rows of the null half read the conditional half's attention output
        if (Mc > 0) next_gemm(W.wq_c, Mc, QD, D, 4); else next_gemm(W.wgu, M, 2 * F, D, 3);
        if (ao_is_mx) rc = gemm_mx(h, nullptr, QD, W.mx_o, h->h, D, M, D, QD, ep, s);
        else if (mx_usable(h, W.mx_o, M, D, QD, 2)) rc = gemm_mx(h, h->ao, QD, W.mx_o, h->h, D, M, D, QD, ep, s);
        else rc = gemm(h, h->ao, QD, W.wo, QD, h->h, D, M, D, QD, ep, s);
        if (rc) return rc;

        // ---- cross attention (base.py:515-526): un-modulated norm, plain residual, cached K/V, no RoPE
        if (Nc > 0) {
        static int mx_cross = -1;
        if (mx_cross < 0) { const char* e = getenv("ACE355_MX_CROSS"); mx_cross = e ? atoi(e) : 0; }
        const bool mx_qc = mx_cross && mx_usable(h, W.mx_qc, Mc, QD, D, 4, QD, QD) && D == 2048;
        if (fold) rc = 0;
        else if (mx_qc) rc = launch_rmsnorm_gs_mx(h->h, W.n_ca, nullptr, h->xq, h->xs, h->xs_pad, Mc, D, eps, 0, S, s);  // (x rstd) w: rmsnorm_mod's product
        else rc = launch_rmsnorm_mod(h->h, W.n_ca, h->xn, Mc, D, eps, nullptr, nullptr, nullptr, nullptr, 0, S, s);
        if (rc) return rc;
        ep = GemmEpilogue{4, nullptr, nullptr, nullptr, 0, S};  // q head-norm in the epilogue (no RoPE on the cross path)
        ep.hn_wq = ep.hn_wk = W.qn_c, ep.hn_cos = ep.hn_sin = nullptr;
        ep.hn_q_cols = ep.hn_qk_cols = QD, ep.hn_eps = eps;
        if (fold) { ep.nc_rowsq = rowsq(li, 1); ep.nc_bias = nullptr; ep.nc_inv_d = inv_d; ep.nc_eps = eps; }
        next_gemm(W.wo_c, Mc, D, QD, 2);
        if (mx_qc) rc = gemm_mx(h, nullptr, D, W.mx_qc, h->qkv, QD, Mc, QD, D, ep, s);
        else rc = gemm(h, h->xn, D, W.wq_c, D, h->qkv, QD, Mc, QD, D, ep, s);
        if (rc) return rc;
        bool cao_is_mx = false;
        {
            AttnArgs a{};
            a.q = h->qkv; a.q_seq_stride = (long)S * QD; a.q_row_stride = QD;
            a.use_tab = 1;
            for (int i = 0; i < Nc; ++i) {
                const CondSlot& cs = h->slots[slots[i]];
                a.k_tab[i] = (unsigned long long)(cs.kv + (size_t)li * cs.L * 2 * KVD);
                a.vt_tab[i] = (unsigned long long)(cs.vt + (size_t)li * h->KVH * 128 * cs.Lpad);
            }
            a.k_head_stride = 128; a.k_row_stride = 2 * KVD;
            a.vt_head_stride = 128L * Lpad; a.vt_ld = Lpad;
            a.out = h->ao; a.o_seq_stride = (long)S * QD; a.o_row_stride = QD;
            a.N = Nc; a.Sq = S; a.Skv = L; a.Hq = h->HQ; a.Hkv = h->KVH; a.window = -1; a.scale = scale;
            a.part = h->attn_part; a.part_floats = h->attn_part_floats;
            a.cu_slots = h->cu_slots;
            if (mx_cross && mx_usable(h, W.mx_oc, Mc, D, QD, 2) && attention_mx_out_ok(a)) {
                a.out_q = h->xq; a.out_scales = h->xs; a.out_pad = h->xs_pad;
                cao_is_mx = true;
            }
            EvScope ev(h, &h->attn_ev, s);
            if (h->profile) h->attn_flops += 4.0 * Nc * h->HQ * (double)S * L * 128.0;
            rc = launch_attention(a, s);
            if (rc) return rc;
        }
        ep = GemmEpilogue{2, nullptr, nullptr, nullptr, 0, S};
        if (fold) {  // the conditional rows' MLP-norm operand
            ep.nf_xg = h->xn; ep.nf_ldx = D; ep.nf_split = Mc;
            ep.nf_gA = ep.nf_gB = g_mlp; ep.nf_sqA = ep.nf_sqB = rowsq(li, 2);
        }
        next_gemm(W.wgu, M, 2 * F, D, 3);
        if (cao_is_mx) rc = gemm_mx(h, nullptr, QD, W.mx_oc, h->h, D, Mc, D, QD, ep, s);
        else if (mx_cross && mx_usable(h, W.mx_oc, Mc, D, QD, 2)) rc = gemm_mx(h, h->ao, QD, W.mx_oc, h->h, D, Mc, D, QD, ep, s);
        else rc = gemm(h, h->ao, QD, W.wo_c, QD, h->h, D, Mc, D, QD, ep, s);
        if (rc) return rc;
        }

        // ---- SwiGLU MLP (base.py:530-533)
        const bool mx_gu = mx_usable(h, W.mx_gu, M, 2 * F, D, 3) && D == 2048;
        if (fold) rc = 0;
        else if (mx_gu)
            rc = launch_rmsnorm_gs_mx(h->h, gs_p + (size_t)(li * 2 + 1) * 2 * D, gs_p + (size_t)(li * 2 + 1) * 2 * D + D, h->xq, h->xs, h->xs_pad,
                                      M, D, eps, gs_stride, S, s);
        else
        rc = launch_rmsnorm_gs(h->h, gs_p + (size_t)(li * 2 + 1) * 2 * D, gs_p + (size_t)(li * 2 + 1) * 2 * D + D, h->xn, M, D, eps,
                               gs_stride, S, s);
        if (rc) return rc;
        ep = GemmEpilogue{3, nullptr, nullptr, nullptr, 0, 0};
        const bool mx_down = mx_usable(h, W.mx_down, M, D, F, 2);
        const bool act_q = mx_gu && mx_down && F % 128 == 0;  // the SwiGLU epilogue writes the down projection's MXFP8 operand itself
        if (act_q) { ep.mxo_scales = h->as_; ep.mxo_pad = h->xs_pad; }
        if (fold) {
            ep.nc_rowsq = rowsq(li, 2); ep.nc_bias = nf.bias_gu + ((size_t)li * nf.rows + nf.step) * 2 * F;
            ep.nc_inv_d = inv_d; ep.nc_eps = eps;
        }
        next_gemm(W.wdown, M, D, F, 2);
        if (mx_gu) rc = gemm_mx(h, nullptr, D, W.mx_gu, act_q ? (void*)h->aq : (void*)h->act, F, M, 2 * F, D, ep, s);
        else if (mx_usable(h, W.mx_gu, M, 2 * F, D, 3)) rc = gemm_mx(h, h->xn, D, W.mx_gu, h->act, F, M, 2 * F, D, ep, s);
        else rc = gemm(h, h->xn, D, W.wgu, D, h->act, F, M, 2 * F, D, ep, s);
        if (rc) return rc;
        ep = GemmEpilogue{2, nullptr, W.sst + 5 * D, tproj_p + 5 * D, tstride, S};
        if (fold && li + 1 < h->NL) {  // the next layer's self-attention norm operand
            ep.nf_xg = h->xn; ep.nf_ldx = D; ep.nf_split = M;
            ep.nf_gA = ep.nf_gB = gs_p + (size_t)((li + 1) * 2 + 0) * 2 * D; ep.nf_sqA = ep.nf_sqB = rowsq(li + 1, 0);
        }
        next_gemm(h->layers[(li + 1) % h->NL].wqkv, M, QKV, D, 4);
        if (act_q) rc = gemm_mx(h, nullptr, F, W.mx_down, h->h, D, M, D, F, ep, s, h->aq, h->as_);
        else if (mx_down) rc = gemm_mx(h, h->act, F, W.mx_down, h->h, D, M, D, F, ep, s);
        else rc = gemm(h, h->act, F, W.wdown, F, h->h, D, M, D, F, ep, s);
        if (rc) return rc;
        if (h->tap_dst[li]) ACE_HIP(hipMemcpyAsync(h->tap_dst[li], h->h, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    }

    // output norm + modulation with temb (base.py:1491-1496): shift = sst[0] + temb, scale = sst[1] + temb
    rc = launch_rmsnorm_mod(h->h, h->norm_out, h->xn, M, D, eps, h->sst_out + D, temb_p, h->sst_out, temb_p,
                            temb_rows == 1 ? 0 : D, S, s);
    if (rc) return rc;
    // proj_out: ConvTranspose1d(D -> 64, k=2, s=2) == GEMM to [M, 128] == [N, 2S, 64] (base.py:1498)
    ep = GemmEpilogue{1, h->b_out, nullptr, nullptr, 0, 0};
    rc = gemm(h, h->xn, D, h->w_out, D, h->vpad, 2 * h->OUTC, M, 2 * h->OUTC, D, ep, s);
    return rc;
}

// The sampling loop of generate_audio (base.py:1913-1981) as a launch sequence: steps x {timestep embedding, decoder forward, guidance +
// update}.  Pure enqueue (no synchronisation, no allocation): runs eagerly or under stream capture.  One SamplerChain = one context (the
// handle, or the chain-2 context of the dual-chain sampler) advancing ITS songs on ITS stream; the chains of a call are stepped in turn
// by the caller, so that the host feeds both hardware queues step by step instead of one chain's whole schedule first.
struct SamplerChain {
    ace355_dit* h = nullptr;
    ace355_sample_params p;      // this chain's view of the request (per-item pointers advanced to its first song)
    int B = 0, T = 0;
    int sde_B = 0;               // songs of the WHOLE request: the per-step stride of sde_noise_dev [steps][B][T][64]
    hipStream_t s = nullptr;
    int cond = 0;
    const int32_t* cond_tab = nullptr;
    bool switched = false;
    int apg_calls = 0;
};
int sampler_begin(SamplerChain& c) {
    const bool do_cfg = c.p.guidance_scale > 1.0f;
    c.cond_tab = c.p.cond_slots_host;  // per-item conditions (NULL: cond_slot for every item)
    c.cond = c.p.cond_slot;
    c.switched = false;
    c.apg_calls = 0;
    return normfold_prepare(c.h, &c.p, c.B * (do_cfg ? 2 : 1), c.T, c.s);
}
int sampler_step(SamplerChain& c, int i, hipEvent_t ev) {
    ace355_dit* h = c.h;
    const ace355_sample_params* p = &c.p;
    hipStream_t s = c.s;
    const int B = c.B, T = c.T;
    const bool do_cfg = p->guidance_scale > 1.0f;
    const int copies = do_cfg ? 2 : 1, N = B * copies;
    const int Tpad = 2 * ((T + 1) / 2);
    int rc;
    int slots[ACE355_MAX_SEQS];
    h->nf.step = i;
    if (i >= p->cover_switch_step && !c.switched) {  // base.py:1916-1927
        c.switched = true;
        ACE_CHECK(p->ctx_non_cover_dev != nullptr, "dit_sample: cover switch needs ctx_non_cover_dev");
        c.cond = p->non_cover_slot;
        c.cond_tab = p->non_cover_slots_host;
        rc = launch_set_xin_ctx(p->ctx_non_cover_dev, h->xin, B, copies, T, Tpad, s);
        if (rc) return rc;
    }
    for (int b = 0; b < B; ++b) {
        slots[b] = c.cond_tab ? c.cond_tab[b] : c.cond;
        if (do_cfg) slots[B + b] = p->null_slot;
    }
    const float t_curr = p->t_sched_host[i], t_prev = p->t_sched_host[i + 1];
    RoctxRange r_step("ace355.sampler_step");
    if (!h->nf.emb_on) {   // (normfold_prepare already computed the embeddings of the whole schedule: forward_core reads row i)
        rc = time_embed(h, &t_curr, &t_curr, 1, s);
        if (rc) return rc;
    }
    // (both copies of a song were filled from the same latents and the same context rows: launch_set_xin_latent / launch_set_xin_ctx)
    h->dup_half = do_cfg;
    rc = forward_core(h, N, T, slots, 1, s);
    h->dup_half = false;
    if (rc) return rc;
    const int apply = (t_curr >= p->cfg_interval_start && t_curr <= p->cfg_interval_end) ? 1 : 0;
    const float dt = t_curr - t_prev;
    StepUpdate up{nullptr, t_curr, 0.f};
    if (p->infer_method == 1) {
        up.sde_noise = p->sde_noise_dev + (size_t)i * c.sde_B * T * h->OUTC;
        // base / sft: linear level (base.py:1972); turbo: the next table value (turbo.py:1980-1984)
        up.t_next = p->sde_next_from_sched ? t_prev : 1.0f - (float)(i + 1) / (float)p->num_steps;
    }
    if (do_cfg && apply && p->use_adg)
        rc = launch_adg_step(h->vpad, (long)B * Tpad * h->OUTC, h->xt, h->xin, copies, B, T, Tpad, p->guidance_scale, t_curr, dt, up, s);
    else
        rc = launch_apg_euler(h->vpad, (long)B * Tpad * h->OUTC, h->avg, h->xt, h->xin, copies, B, T, Tpad, p->guidance_scale, dt,
                              apply, do_cfg ? 1 : 0, c.apg_calls == 0 ? 1 : 0, up, s);
    if (rc) return rc;
    if (do_cfg && apply && !p->use_adg) ++c.apg_calls;
    if (ev) hipEventRecord(ev, s);
    return 0;
}

// One or two chains of one call, stepped in turn.  Two chains: chain 2 (the context) runs on the handle's side stream between a fork
// and a join event on `s` - the same sequence eagerly and under stream capture (the side stream joins the capture through the event).
int run_sampler_chains(ace355_dit* h, SamplerChain* chains, int nchains, hipStream_t s, std::vector<hipEvent_t>* evs) {
    int rc = 0;
    if (nchains == 2) {
        ACE_HIP(hipEventRecord(h->fk.ev_fork, s));
        ACE_HIP(hipStreamWaitEvent(chains[1].s, h->fk.ev_fork, 0));
    }
    for (int k = nchains - 1; k >= 0 && !rc; --k) rc = sampler_begin(chains[k]);
    const int steps = chains[0].p.num_steps;
    for (int i = 0; i < steps && !rc; ++i)
        for (int k = nchains - 1; k >= 0 && !rc; --k) rc = sampler_step(chains[k], i, (k == 0 && evs) ? (*evs)[i + 1] : nullptr);
    if (nchains == 2) {
        // the join is recorded on EVERY path (advisor r4): after a failed step the side stream may still hold queued work on the chain-2
        // context's buffers, and the caller only ever synchronises `s` before it frees or regrows them
        const hipError_t e1 = hipEventRecord(h->fk.ev_join, chains[1].s);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, h->fk.ev_join, 0) : e1;
        if (e2 != hipSuccess) {
            (void)hipStreamSynchronize(chains[1].s);   // (outside a capture this still orders the side stream's work before the caller's next step)
            if (!rc) return hip_fail(e2, "dual-chain join", __FILE__, __LINE__);
        }
    }
    return rc;
}

// ---- chain-2 context of the dual-chain sampler
int chain_ctx_sync(ace355_dit* h) {
    if (!h->dual.ctx) {
        ace355_dit* c = new ace355_dit();
        c->alias = true;
        c->dual.mode = 0;
        ALLOC(c->allocs, c->flags_dev, 4);
        ALLOC(c->allocs, c->sk_cnt, SK_CNT_INTS);
        ACE_HIP(hipMemset(c->sk_cnt, 0, SK_CNT_INTS * sizeof(int)));
        ALLOC(c->allocs, c->fk.sk_cnt, SK_CNT_INTS);
        ACE_HIP(hipMemset(c->fk.sk_cnt, 0, SK_CNT_INTS * sizeof(int)));
        c->attn_part_floats = h->attn_part_floats;
        ALLOC(c->allocs, c->attn_part, (size_t)c->attn_part_floats);
        h->dual.ctx = c;
    }
    ace355_dit* c = h->dual.ctx;
    // everything the forward reads but does not own: dimensions, packed weights (+ their MXFP8 copies), norm tables, condition slots, rope
    c->cfg = h->cfg;
    c->D = h->D; c->F = h->F; c->QD = h->QD; c->KVD = h->KVD; c->NL = h->NL; c->KVH = h->KVH; c->HQ = h->HQ; c->OUTC = h->OUTC;
    c->layers = h->layers;
    c->w_in = h->w_in; c->w_out = h->w_out; c->w_cond = h->w_cond;
    c->b_in = h->b_in; c->b_out = h->b_out; c->b_cond = h->b_cond; c->norm_out = h->norm_out; c->sst_out = h->sst_out;
    c->te[0] = h->te[0]; c->te[1] = h->te[1];
    c->mod_tab = h->mod_tab;
    c->finalized = h->finalized;
    for (int i = 0; i < ACE355_MAX_SLOTS; ++i) c->slots[i] = h->slots[i];
    c->rope_cos = h->rope_cos; c->rope_sin = h->rope_sin; c->rope_S = h->rope_S;
    if (c->precision != h->precision || c->weights_fp8wo != h->weights_fp8wo || c->nf.enabled != h->nf.enabled) c->nf.key.clear();
    c->precision = h->precision; c->weights_fp8wo = h->weights_fp8wo; c->mx_min_rows = h->mx_min_rows;
    c->nf.enabled = h->nf.enabled; c->nf.min_rows = h->nf.min_rows;
    c->profile = h->profile;
    c->dedup_on = h->dedup_on;
    return 0;
}
void chain_ctx_destroy(ace355_dit* c) {
    if (!c) return;
    for (void* p : c->allocs) hipFree(p);
    for (void* p : c->ws_allocs) hipFree(p);
    for (void* p : c->nf.allocs) hipFree(p);
    for (auto& e : c->gemm_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : c->attn_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    delete c;
}

// Do the caller's stream and the side stream sit on different hardware queues?  The runtime hands out a few queues (GPU_MAX_HW_QUEUES,
// 4 by default) and shares them between streams beyond that; two streams on one queue run their kernels strictly one after the
// other (the two-stream probes of rounds 2 / 3 measured exactly that and read it as "no gain").  Two 60 us spin kernels, one per stream,
// between two events: ~60 us side by side, ~120 us in series.  A serialised side stream is replaced by a fresh one (up to 6 tries).
// Once per caller stream; synchronises it.  Not under capture.
int dual_probe_streams(ace355_dit* h, hipStream_t s) {
    for (const auto& pr : h->dual.probed)
        if (pr.first == s) { h->dual.concurrent = pr.second; return 0; }
    h->dual.concurrent = false;
    {   // a caller that is capturing `s` into a graph of its own cannot be synchronised: unknown stream = one chain for this call
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return 0; }
        if (cap != hipStreamCaptureStatusNone) return 0;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ACE_HIP(hipEventCreate(&e0));
    ACE_HIP(hipEventCreate(&e1));
    std::vector<hipStream_t> losers;
    int rc = 0;
    for (int attempt = 0; attempt < 6 && !h->dual.concurrent; ++attempt) {
        float best = 1e30f;
        for (int rep = 0; rep < 2 && !rc; ++rep) {   // (the first pair also pays the streams' first-use set-up)
            hipEventRecord(e0, s);
            hipEventRecord(h->fk.ev_fork, s);
            hipStreamWaitEvent(h->fk.side, h->fk.ev_fork, 0);
            rc = launch_spin(60, h->fk.side);
            hipEventRecord(h->fk.ev_join, h->fk.side);
            if (!rc) rc = launch_spin(60, s);
            hipStreamWaitEvent(s, h->fk.ev_join, 0);
            hipEventRecord(e1, s);
            if (hipEventSynchronize(e1) != hipSuccess) rc = 2;
            float ms = 0.f;
            if (!rc && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) best = std::min(best, ms);
        }
        if (rc) break;
        if (best < 0.095f) { h->dual.concurrent = true; break; }
        losers.push_back(h->fk.side);   // (kept alive until a winner is found: a destroyed stream's queue slot would be handed out again)
        h->fk.side = nullptr;
        h->dual.probed.clear();         // (answers about the old side stream)
        if (hipStreamCreateWithFlags(&h->fk.side, hipStreamNonBlocking) != hipSuccess) { h->fk.side = losers.back(); losers.pop_back(); break; }
    }
    for (hipStream_t l : losers) hipStreamDestroy(l);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (!rc) {
        if (h->dual.probed.size() >= 16) h->dual.probed.erase(h->dual.probed.begin());
        h->dual.probed.push_back({s, h->dual.concurrent});
    }
    if (!h->dual.concurrent && !rc) {
        static bool said = false;
        if (!said) fprintf(stderr, "[ace355] no side stream on a hardware queue of its own for this caller stream: the sampler runs as one chain\n");
        said = true;
    }
    return rc;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

int ace355_dit_create(const ace355_dit_config* cfg, ace355_dit** out) {
    ACE_CHECK(cfg && out, "dit_create: null argument");
    ACE_CHECK(cfg->head_dim == 128, "dit_create: head_dim must be 128");
    ACE_CHECK(cfg->patch_size == 2, "dit_create: patch_size must be 2");
    ACE_CHECK(cfg->hidden_size % 256 == 0 || cfg->hidden_size % 64 == 0, "dit_create: hidden_size % 64");
    ACE_CHECK(cfg->hidden_size % 64 == 0 && cfg->intermediate_size % 64 == 0, "dit_create: sizes must be multiples of 64");
    ACE_CHECK((2 * cfg->in_channels) % 64 == 0, "dit_create: 2*in_channels must be a multiple of 64");
    ACE_CHECK(cfg->num_heads % cfg->num_kv_heads == 0 && cfg->num_layers > 0 && cfg->num_layers <= 64, "dit_create: heads/layers");
    ACE_CHECK(cfg->out_channels % 4 == 0 && cfg->out_channels == 64, "dit_create: out_channels must be 64");
    ace355_dit* h = new ace355_dit();
    h->cfg = *cfg;
    h->D = cfg->hidden_size; h->F = cfg->intermediate_size; h->NL = cfg->num_layers;
    h->HQ = cfg->num_heads; h->KVH = cfg->num_kv_heads; h->QD = cfg->num_heads * 128; h->KVD = cfg->num_kv_heads * 128;
    h->OUTC = cfg->out_channels;
    const size_t D = h->D, F = h->F, QD = h->QD, KVD = h->KVD;
    h->layers.resize(h->NL);
    for (LayerW& L : h->layers) {
        ALLOC(h->allocs, L.wqkv, (QD + 2 * KVD) * D);
        ALLOC(h->allocs, L.wo, D * QD);
        ALLOC(h->allocs, L.wq_c, QD * D);
        ALLOC(h->allocs, L.wkv_c, 2 * KVD * D);
        ALLOC(h->allocs, L.wo_c, D * QD);
        ALLOC(h->allocs, L.wgu, 2 * F * D);
        ALLOC(h->allocs, L.wdown, D * F);
        ALLOC(h->allocs, L.n_sa, D); ALLOC(h->allocs, L.n_ca, D); ALLOC(h->allocs, L.n_mlp, D);
        ALLOC(h->allocs, L.qn_s, 128); ALLOC(h->allocs, L.kn_s, 128); ALLOC(h->allocs, L.qn_c, 128); ALLOC(h->allocs, L.kn_c, 128);
        ALLOC(h->allocs, L.sst, 6 * D);
    }
    const size_t CP = (size_t)cfg->in_channels * cfg->patch_size;
    ALLOC(h->allocs, h->w_in, D * CP);
    ALLOC(h->allocs, h->b_in, D);
    ALLOC(h->allocs, h->w_out, (size_t)2 * h->OUTC * D);
    ALLOC(h->allocs, h->b_out, (size_t)2 * h->OUTC);
    for (int e = 0; e < 2; ++e) {
        ALLOC(h->allocs, h->te[e].l1, D * 256); ALLOC(h->allocs, h->te[e].b1, D);
        ALLOC(h->allocs, h->te[e].l2, D * D); ALLOC(h->allocs, h->te[e].b2, D);
        ALLOC(h->allocs, h->te[e].tp, 6 * D * D); ALLOC(h->allocs, h->te[e].bp, 6 * D);
    }
    ALLOC(h->allocs, h->w_cond, D * D);
    ALLOC(h->allocs, h->b_cond, D);
    ALLOC(h->allocs, h->norm_out, D);
    ALLOC(h->allocs, h->sst_out, 2 * D);
    ALLOC(h->allocs, h->flags_dev, 4);
    ALLOC(h->allocs, h->sk_cnt, SK_CNT_INTS);
    ALLOC(h->allocs, h->fk.sk_cnt, SK_CNT_INTS);
    ACE_HIP(hipMemset(h->fk.sk_cnt, 0, SK_CNT_INTS * sizeof(int)));
    h->attn_part_floats = 16L << 20;   // 64 MB: 8 parts of a 2 x 16 x 375-row problem (12.7 M floats); larger problems do not split
    ALLOC(h->allocs, h->attn_part, (size_t)h->attn_part_floats);
    ACE_HIP(hipMemset(h->sk_cnt, 0, SK_CNT_INTS * sizeof(int)));
    if (int prc = gemm_verify_splitk_placement()) return prc;
    h->expected_tensors = (size_t)h->NL * 19 + 4 + 12 + 4;
    if (const char* e = getenv("ACE355_SAMPLE_GRAPH")) h->graph_mode = atoi(e) != 0;
    if (const char* e = getenv("ACE355_DEDUP0")) h->dedup_on = atoi(e) != 0;
    if (const char* e = getenv("ACE355_DUAL")) h->dual.mode = atoi(e);
    if (const char* e = getenv("ACE355_DUAL_SLOTS_MIN_ROWS")) h->dual.slots_min_rows = atoi(e);
    if (const char* e = getenv("ACE355_DUAL_MAX_ROWS")) h->dual.max_rows = atoi(e);
    {
        ACE_HIP(hipStreamCreateWithFlags(&h->fk.side, hipStreamNonBlocking));
        ACE_HIP(hipEventCreateWithFlags(&h->fk.ev_fork, hipEventDisableTiming));
        ACE_HIP(hipEventCreateWithFlags(&h->fk.ev_join, hipEventDisableTiming));
    }
    if (const char* e = getenv("ACE355_NORM_FOLD")) h->nf.enabled = atoi(e);
    if (const char* e = getenv("ACE355_NORM_FOLD_MIN_ROWS")) h->nf.min_rows = atoi(e);
    *out = h;
    return ACE355_OK;
}

void ace355_dit_destroy(ace355_dit* h) {
    if (!h) return;
    hipDeviceSynchronize();
    chain_ctx_destroy(h->dual.ctx);
    for (void* p : h->allocs) hipFree(p);
    for (void* p : h->ws_allocs) hipFree(p);
    for (void* p : h->mx_allocs) hipFree(p);
    for (void* p : h->nf.allocs) hipFree(p);
    for (CondSlot& c : h->slots) {
        if (c.kv) hipFree(c.kv);
        if (c.vt) hipFree(c.vt);
        if (c.cross_const) hipFree(c.cross_const);
    }
    if (h->g_ctx_nc) hipFree(h->g_ctx_nc);
    if (h->g_sde) hipFree(h->g_sde);
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    if (h->fk.side) hipStreamDestroy(h->fk.side);
    if (h->fk.ev_fork) hipEventDestroy(h->fk.ev_fork);
    if (h->fk.ev_join) hipEventDestroy(h->fk.ev_join);
    if (h->graph_stream) hipStreamDestroy(h->graph_stream);
    if (h->graph_in) hipEventDestroy(h->graph_in);
    if (h->graph_out) hipEventDestroy(h->graph_out);
    if (h->stage) hipFree(h->stage);
    if (h->rope_cos) hipFree(h->rope_cos);
    if (h->rope_sin) hipFree(h->rope_sin);
    if (h->enc_bf) hipFree(h->enc_bf);
    if (h->enc_emb) hipFree(h->enc_emb);
    for (auto& e : h->gemm_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    for (auto& e : h->attn_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    delete h;
}

int ace355_dit_load_tensor(ace355_dit* h, const char* name, const void* data, int dtype, int64_t numel, int is_device) {
    ACE_CHECK(h && name && data, "dit_load_tensor: null argument");
    ACE_CHECK(dtype == ACE355_DTYPE_F32 || dtype == ACE355_DTYPE_BF16, "dit_load_tensor: dtype");
    Dest d;
    if (!resolve(h, name, &d)) {
        set_error(std::string("dit_load_tensor: unknown tensor name '") + name + "'");
        return ACE355_ERR_INVALID;
    }
    if (numel != d.rows * d.cols) {
        set_error(std::string("dit_load_tensor: wrong element count for '") + name + "': got " + std::to_string(numel) +
                  ", expected " + std::to_string(d.rows * d.cols));
        return ACE355_ERR_INVALID;
    }
    const size_t esz = dtype == ACE355_DTYPE_F32 ? 4 : 2;
    const void* src = data;
    if (!is_device) {
        const size_t bytes = (size_t)numel * esz;
        if (bytes > h->stage_bytes) {
            if (h->stage) ACE_HIP(hipFree(h->stage));
            h->stage = nullptr;
            ACE_HIP(hipMalloc(&h->stage, bytes));
            h->stage_bytes = bytes;
        }
        ACE_HIP(hipMemcpy(h->stage, data, bytes, hipMemcpyHostToDevice));
        src = h->stage;
    }
    int rc = launch_pack(src, dtype, d.dst, d.is_bf16, d.mode, d.rows, d.cols, d.dst_ld, d.dst_row0, d.p0, d.p1, nullptr);
    if (rc) return rc;
    if (std::string(name) == "proj_out.1.bias") {  // bias[p*64 + c] = b[c]
        rc = launch_pack(src, dtype, h->b_out, 0, PACK_ROWS, 1, h->OUTC, h->OUTC, 0, 0, 0, nullptr);
        if (rc) return rc;
        rc = launch_pack(src, dtype, h->b_out + h->OUTC, 0, PACK_ROWS, 1, h->OUTC, h->OUTC, 0, 0, 0, nullptr);
        if (rc) return rc;
    }
    ACE_HIP(hipDeviceSynchronize());
    h->loaded.insert(name);
    h->finalized = false;
    h->weights_fp8wo = false;   // (fresh weights: the round trip, if wanted, is applied again by set_precision)
    return ACE355_OK;
}

int ace355_dit_finalize(ace355_dit* h) {
    ACE_CHECK(h, "dit_finalize: null handle");
    if (h->loaded.size() != h->expected_tensors) {
        set_error("dit_finalize: " + std::to_string(h->loaded.size()) + " of " + std::to_string(h->expected_tensors) +
                  " tensors loaded");
        return ACE355_ERR_STATE;
    }
    if (h->stage) { hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
    {   // modulated norms: scale_shift_table rows (shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate), base.py:492-497
        std::vector<ModEntry> tab;
        for (const LayerW& L : h->layers) {
            tab.push_back(ModEntry{L.n_sa, L.sst + 1 * h->D, L.sst + 0 * h->D, 1L * h->D, 0L * h->D});
            tab.push_back(ModEntry{L.n_mlp, L.sst + 4 * h->D, L.sst + 3 * h->D, 4L * h->D, 3L * h->D});
        }
        if (!h->mod_tab) ALLOC(h->allocs, h->mod_tab, tab.size());
        ACE_HIP(hipMemcpy(h->mod_tab, tab.data(), tab.size() * sizeof(ModEntry), hipMemcpyHostToDevice));
    }
    h->nf.key.clear();   // embedding / bias tables are functions of the (possibly new) weights
    if (h->dual.ctx) h->dual.ctx->nf.key.clear();
    h->finalized = true;
    return ACE355_OK;
}

int ace355_dit_set_condition(ace355_dit* h, int slot, const float* enc_dev, int rows, int L, void* stream) {
    ACE_CHECK(h && enc_dev, "set_condition: null argument");
    if (!h->finalized) { set_error("set_condition: call ace355_dit_finalize first"); return ACE355_ERR_STATE; }
    ACE_CHECK(slot >= 0 && slot < ACE355_MAX_SLOTS, "set_condition: slot out of range");
    ACE_CHECK(L > 0 && (rows == L || rows == 1), "set_condition: rows must be L or 1");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->D, KVD = h->KVD;
    const int Lpad = ((L + 63) / 64) * 64;
    if (L > h->enc_cap) {
        ACE_HIP(hipStreamSynchronize(s));
        if (h->enc_bf) hipFree(h->enc_bf);
        if (h->enc_emb) hipFree(h->enc_emb);
        ACE_HIP(hipMalloc((void**)&h->enc_bf, ((size_t)L * D + h->QD) * 2 + 256));
        ACE_HIP(hipMalloc((void**)&h->enc_emb, (size_t)2 * L * D * 2 + 256));
        h->enc_cap = L;
    }
    CondSlot& cs = h->slots[slot];
    if (L > cs.cap) {
        ACE_HIP(hipStreamSynchronize(s));
        if (cs.kv) hipFree(cs.kv);
        if (cs.vt) hipFree(cs.vt);
        ACE_HIP(hipMalloc((void**)&cs.kv, (size_t)h->NL * L * 2 * KVD * 2 + 256));
        ACE_HIP(hipMalloc((void**)&cs.vt, (size_t)h->NL * h->KVH * 128 * Lpad * 2 + 256));
        cs.cap = L;
        h->cond_epoch++;
    }
    if (!cs.cross_const) {
        ACE_HIP(hipMalloc((void**)&cs.cross_const, (size_t)h->NL * D * 4 + 256));
        h->cond_epoch++;
    }
    cs.valid = false;
    int rc = launch_f32_to_bf16(enc_dev, h->enc_bf, (long)rows * D, s);
    if (rc) return rc;
    // condition_embedder (base.py:1359)
    GemmEpilogue ep{0, h->b_cond, nullptr, nullptr, 0, 0};
    bf16_t* emb = h->enc_emb;
    rc = gemm(h, h->enc_bf, D, h->w_cond, D, emb, D, rows, D, D, ep, s);
    if (rc) return rc;
    if (rows == 1 && L > 1) {  // null_condition_emb.expand_as(enc) (base.py:1907)
        bf16_t* wide = h->enc_emb + (size_t)L * D;
        rc = launch_bcast_rows(emb, wide, L, D, s);
        if (rc) return rc;
        emb = wide;
    }
    for (int li = 0; li < h->NL; ++li) {
        const LayerW& W = h->layers[li];
        bf16_t* kv = cs.kv + (size_t)li * L * 2 * KVD;
        ep = GemmEpilogue{0, nullptr, nullptr, nullptr, 0, 0};
        rc = gemm(h, emb, D, W.wkv_c, D, kv, 2 * KVD, L, 2 * KVD, D, ep, s);  // K | V (base.py:320-321)
        if (rc) return rc;
        rc = launch_headnorm_rope(kv, L, 2 * KVD, 0, h->KVH, W.kn_c, h->cfg.rms_norm_eps, nullptr, nullptr, L, s);
        if (rc) return rc;
        rc = launch_transpose_v(kv, 2 * KVD, KVD, 1, L, h->KVH, cs.vt + (size_t)li * h->KVH * 128 * Lpad, Lpad, s);
        if (rc) return rc;
        if (rows == 1) {
            // softmax over L identical keys is uniform, so cross-attention returns V's (single) row for every query:
            // the layer's cross-attention residual is the constant o_proj(expand_heads(v)) ("degenerate null branch")
            rc = launch_expand_kv_heads(kv + KVD, h->enc_bf, h->HQ, h->KVH, s);  // enc_bf is free again (>= D >= QD elements? see check)
            if (rc) return rc;
            ep = GemmEpilogue{1, nullptr, nullptr, nullptr, 0, 0};
            rc = gemm(h, h->enc_bf, h->QD, W.wo_c, h->QD, cs.cross_const + (size_t)li * D, D, 1, D, h->QD, ep, s);
            if (rc) return rc;
        }
    }
    if (cs.broadcast != (rows == 1) || cs.L != L) h->cond_epoch++;  // launch shapes of later forwards change
    cs.broadcast = rows == 1;
    cs.L = L;
    cs.Lpad = Lpad;
    cs.valid = true;
    return ACE355_OK;
}

int ace355_dit_forward(ace355_dit* h, const float* x_dev, const float* ctx_dev, const float* t_host, const float* t_r_host,
                       const int32_t* slots_host, int N, int T, float* v_out_dev, void* stream) {
    ACE_CHECK(h && x_dev && ctx_dev && t_host && t_r_host && slots_host && v_out_dev, "dit_forward: null argument");
    if (!h->finalized) { set_error("dit_forward: call ace355_dit_finalize first"); return ACE355_ERR_STATE; }
    ACE_CHECK(N > 0 && N <= ACE355_MAX_SEQS && T > 0, "dit_forward: N in [1,64], T > 0");
    hipStream_t s = (hipStream_t)stream;
    h->vt_key_N = h->vt_key_S = -1;   // (see ace355_dit_sample)
    h->cu_slots = 0;
    int rc = ensure_workspace(h, N, T, s);
    if (rc) return rc;
    const int Tpad = 2 * ((T + 1) / 2);
    rc = launch_pack_xin(x_dev, ctx_dev, h->xin, N, T, Tpad, s);
    if (rc) return rc;
    rc = time_embed(h, t_host, t_r_host, N, s);
    if (rc) return rc;
    h->nf.on = false;   // the folded-norm bias tables exist per sampler schedule only
    h->nf.emb_on = false;
    rc = forward_core(h, N, T, slots_host, N, s);
    if (rc) return rc;
    return launch_copy_v(h->vpad, v_out_dev, N, T, Tpad, s);
}

int ace355_dit_sample(ace355_dit* h, const float* xt0_dev, const float* ctx_dev, int B, int T, const ace355_sample_params* p,
                      float* latents_out_dev, float* per_step_ms_host, void* stream) {
    ACE_CHECK(h && xt0_dev && ctx_dev && p && latents_out_dev && p->t_sched_host, "dit_sample: null argument");
    if (!h->finalized) { set_error("dit_sample: call ace355_dit_finalize first"); return ACE355_ERR_STATE; }
    ACE_CHECK(B > 0 && T > 0 && p->num_steps > 0, "dit_sample: empty problem");
    ACE_CHECK(p->infer_method == 0 || p->infer_method == 1, "dit_sample: infer_method must be 0 (ode) or 1 (sde)");
    ACE_CHECK(p->infer_method == 0 || p->sde_noise_dev != nullptr, "dit_sample: infer_method sde needs sde_noise_dev [steps,B,T,64]");
    const bool do_cfg = p->guidance_scale > 1.0f;
    const int copies = do_cfg ? 2 : 1, N = B * copies;
    ACE_CHECK(N <= ACE355_MAX_SEQS, "dit_sample: at most 64 sequences per call");
    hipStream_t s = (hipStream_t)stream;
    RoctxRange r_sample("ace355.dit_sample");
    const int S = (T + 1) / 2;
    const int Tpad = 2 * S;
    const int OC = h->OUTC;

    // ---- one chain or two (dual-chain sampler, see ace355_dit::Dual): songs [0, B0) stay on this handle and the caller's stream,
    // songs [B0, B) go to the chain-2 context on the side stream.  Taps address the rows of the whole batch: one chain.
    bool taps = false;
    for (int l = 0; l < h->NL; ++l) taps = taps || h->tap_dst[l] != nullptr;
    const bool graph = h->graph_mode && !per_step_ms_host && !h->profile && !taps;
    if (graph && !h->graph_stream) {
        ACE_HIP(hipStreamCreateWithFlags(&h->graph_stream, hipStreamNonBlocking));
        ACE_HIP(hipEventCreateWithFlags(&h->graph_in, hipEventDisableTiming));
        ACE_HIP(hipEventCreateWithFlags(&h->graph_out, hipEventDisableTiming));
    }
    hipStream_t run_s = graph ? h->graph_stream : s;   // the stream the loop is enqueued (or captured) on
    int nchains = 1;
    // (under graph replay the two chains become two branches of one graph, which the runtime places on streams of its own choosing:
    //  measured 281 ms against 214 ms for the one-chain graph at 2 songs - the default policy keeps a captured call on one chain)
    // (default policy, round 4 late: two chains when ONE chain's launches would leave part of the chip without a tile.  fill = share of the
    //  CU slots of the rounds a one-chain N = 2048 launch needs that hold a tile: 16 column tiles of the 192x128 mid tile up to 3072 rows, 8 of
    //  the 192x256 tile above.  Same-box ABAB, 30 s songs with decode, one chain -> two: 2 songs 198.7 -> 194.5 ms, 3: 266.1 -> 251.0, 4 (fill
    //  1.0): 290.4 -> 303.5, 5 (0.63): 397.0 -> 362.7, 6 (0.75): 430.4 -> 408.9, 7 (0.88): 461.0 -> 457.6 on one box and 451.9 -> 460.5 on
    //  another (hence the 0.85), 8 (1.0): 486.7 -> 518.5, 9 (0.56): 657.6 -> 581.9, 10 (0.63): 681.9 -> 654.0: profiles/r04/r04_dual_policy_quiet_ab.txt)
    auto one_chain_fill = [](long rows) -> double {
        const long tiles = ((rows + 191) / 192) * (rows <= 3072 ? 16 : 8);
        return (double)tiles / (double)(((tiles + 255) / 256) * 256);
    };
    const bool under_filled = (long)N * S <= h->dual.max_rows || one_chain_fill((long)N * S) < 0.85;
    if (h->dual.mode && B >= 2 && !taps && h->fk.side && (h->dual.mode >= 2 || (under_filled && !graph))) {
        int rc0 = dual_probe_streams(h, run_s);
        if (rc0) return rc0;
        if (h->dual.concurrent) nchains = 2;
    }
    const int B0 = nchains == 2 ? (B + 1) / 2 : B, B1 = B - B0;
    ace355_dit* ctxs[2] = {h, nullptr};
    if (nchains == 2) {
        int rc0 = chain_ctx_sync(h);
        if (rc0) return rc0;
        ctxs[1] = h->dual.ctx;
    }
    const int Bc[2] = {B0, B1}, b0[2] = {0, B0};
    struct CallState {   // per-call launch hints of the handle and its chain context: reset on EVERY exit path (advisor r4)
        ace355_dit* h;
        ~CallState() {
            h->cu_slots = 0;
            h->pf_block = false;
            if (h->dual.ctx) { h->dual.ctx->cu_slots = 0; h->dual.ctx->pf_block = false; }
        }
    } call_state{h};
    if (nchains == 2) h->pf_block = h->dual.ctx->pf_block = true;
    int rc;
    for (int k = 0; k < nchains; ++k) {
        ace355_dit* c = ctxs[k];
        // launches planned for half the chip when there are two chains of big launches (small ones never fill a half anyway)
        c->cu_slots = (nchains == 2 && Bc[k] * copies * S >= h->dual.slots_min_rows) ? 128 : 0;
        // the pad columns [S, Sp) of V^T are zeroed by the first forward of EVERY call (12.6 MB at the metric shape), also inside a
        // captured graph: a replay skips forward_core, so a key that survived across calls could describe a buffer another shape had
        // since written (stale - possibly non-finite - V values under zero attention weights; advisor r3)
        c->vt_key_N = c->vt_key_S = -1;
        if (k == 1 && c->rope_S < S) {   // (the handle's rope tables grow in its own ensure_workspace: pick the new ones up)
            c->rope_cos = nullptr; c->rope_sin = nullptr; c->rope_S = 0;
        }
    }
    rc = ensure_workspace(h, B0 * copies, T, s);
    if (rc) return rc;
    rc = normfold_reserve(h, p->num_steps, B0 * copies, T, s);
    if (rc) return rc;
    if (nchains == 2) {
        ace355_dit* c = ctxs[1];
        c->rope_cos = h->rope_cos; c->rope_sin = h->rope_sin; c->rope_S = h->rope_S;   // (ensure_rope above covers S; the context never grows them)
        const long e_before = c->ws_epoch;
        rc = ensure_workspace(c, B1 * copies, T, s);
        if (rc) return rc;
        rc = normfold_reserve(c, p->num_steps, B1 * copies, T, s);
        if (rc) return rc;
        if (c->ws_epoch != e_before) h->ws_epoch++;   // a captured graph holds the context's buffers too
    }
    const size_t item_lat = (size_t)T * OC, item_ctx = (size_t)T * 2 * OC;
    for (int k = 0; k < nchains; ++k) {   // inputs into the chains' own buffers (on the caller's stream, ahead of the fork)
        ace355_dit* c = ctxs[k];
        ACE_HIP(hipMemcpyAsync(c->xt, xt0_dev + b0[k] * item_lat, Bc[k] * item_lat * sizeof(float), hipMemcpyDeviceToDevice, s));
        rc = launch_set_xin_ctx(ctx_dev + b0[k] * item_ctx, c->xin, Bc[k], copies, T, Tpad, s);
        if (rc) return rc;
        rc = launch_set_xin_latent(c->xt, c->xin, Bc[k], copies, T, Tpad, s);
        if (rc) return rc;
    }

    struct Events {  // destroyed on every exit path (a failing step used to leak them)
        std::vector<hipEvent_t> v;
        ~Events() { for (auto& e : v) if (e) hipEventDestroy(e); }
    } evh;
    std::vector<hipEvent_t>& evs = evh.v;
    if (per_step_ms_host) {
        evs.assign(p->num_steps + 1, nullptr);
        for (auto& e : evs) ACE_HIP(hipEventCreate(&e));
        hipEventRecord(evs[0], s);
    }
    // chain views of the request: per-item pointers advanced to the chain's first song
    ace355_sample_params pg = *p;   // graph mode: the loop reads handle-owned copies of the caller's optional tensors
    auto own = [&](const float* src, size_t n, float** dst, size_t* cap) -> int {
        if (!src) return 0;
        if (n > *cap) {
            ACE_HIP(hipStreamSynchronize(s));
            if (*dst) hipFree(*dst);
            *dst = nullptr; *cap = 0;
            ACE_HIP(hipMalloc((void**)dst, n * sizeof(float) + 256));
            *cap = n;
            h->ws_epoch++;
        }
        ACE_HIP(hipMemcpyAsync(*dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    };
    if (graph) {
        if ((rc = own(p->ctx_non_cover_dev, (size_t)B * T * 2 * OC, &h->g_ctx_nc, &h->g_ctx_nc_n))) return rc;
        if ((rc = own(p->sde_noise_dev, (size_t)p->num_steps * B * T * OC, &h->g_sde, &h->g_sde_n))) return rc;
        if (p->ctx_non_cover_dev) pg.ctx_non_cover_dev = h->g_ctx_nc;
        if (p->sde_noise_dev) pg.sde_noise_dev = h->g_sde;
    }
    // (per_step_ms_host, when asked for, times chain 1's steps on the caller's stream: the chains advance side by side, step for step)
    SamplerChain chains[2];
    for (int k = 0; k < nchains; ++k) {
        SamplerChain& c = chains[k];
        c.h = ctxs[k];
        c.p = pg;
        c.B = Bc[k]; c.T = T; c.sde_B = B;
        c.s = k == 0 ? run_s : h->fk.side;
        if (c.p.cond_slots_host) c.p.cond_slots_host += b0[k];
        if (c.p.non_cover_slots_host) c.p.non_cover_slots_host += b0[k];
        if (c.p.ctx_non_cover_dev) c.p.ctx_non_cover_dev += b0[k] * item_ctx;
        if (c.p.sde_noise_dev) c.p.sde_noise_dev += b0[k] * item_lat;
    }
    // hipGraph replay: the loop is a fixed launch sequence for fixed (shapes, schedule, knobs, slot layout, buffers, chain split):
    // captured once, replayed for every later call with the same key (the latent / context inputs were copied into the
    // chains' own buffers above, the result is copied out below, so caller pointers are not part of the graph)
    bool done = false;
    if (graph) {
        std::string key((const char*)p->t_sched_host, (size_t)(p->num_steps + 1) * sizeof(float));
        auto add = [&](const void* q, size_t n) { key.append((const char*)q, n); };
        const long scal[] = {B, T, p->num_steps, p->infer_method, p->use_adg, p->cond_slot, p->null_slot, p->cover_switch_step,
                             p->non_cover_slot, p->sde_next_from_sched, h->ws_epoch, h->cond_epoch,
                             p->ctx_non_cover_dev ? 1L : 0L, p->sde_noise_dev ? 1L : 0L, nchains, (long)(uintptr_t)h->fk.side};
        add(scal, sizeof(scal));
        const float fl[] = {p->guidance_scale, p->cfg_interval_start, p->cfg_interval_end};
        add(fl, sizeof(fl));
        if (p->cond_slots_host) add(p->cond_slots_host, sizeof(int32_t) * B);
        key.push_back('|');
        if (p->non_cover_slots_host) add(p->non_cover_slots_host, sizeof(int32_t) * B);
        hipStream_t gs = h->graph_stream;
        ACE_HIP(hipEventRecord(h->graph_in, s));            // the input copies above
        ACE_HIP(hipStreamWaitEvent(gs, h->graph_in, 0));
        if (h->graph_exec && key == h->graph_key) {
            ACE_HIP(hipGraphLaunch(h->graph_exec, gs));
            h->graph_replays++;
            done = true;
        } else {
            if (h->graph_exec) { hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            if (hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                rc = run_sampler_chains(h, chains, nchains, gs, nullptr);
                const hipError_t e = hipStreamEndCapture(gs, &g);
                if (rc) { if (g) hipGraphDestroy(g); return rc; }
                if (e == hipSuccess && g && hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                    h->graph_key = key;
                    h->graph_captures++;
                    hipGraphDestroy(g);
                    ACE_HIP(hipGraphLaunch(h->graph_exec, gs));
                    done = true;
                } else {
                    if (g) hipGraphDestroy(g);
                    h->graph_exec = nullptr;
                    (void)hipGetLastError();  // capture / instantiate refused: run eagerly below
                }
            } else {
                (void)hipGetLastError();
            }
        }
        // the output copy below (and whatever the caller enqueues next) waits for the replay / for an eager fallback on gs
        if (!done) {
            rc = run_sampler_chains(h, chains, nchains, gs, nullptr);
            if (rc) return rc;
        }
        ACE_HIP(hipEventRecord(h->graph_out, gs));
        ACE_HIP(hipStreamWaitEvent(s, h->graph_out, 0));
    } else {
        rc = run_sampler_chains(h, chains, nchains, s, per_step_ms_host ? &evs : nullptr);
        if (rc) return rc;
    }
    if (nchains == 2) h->dual.calls++;
    h->cu_slots = 0;   // (a host-side launch hint: later single-chain work on this handle plans for the whole chip again)
    for (int k = 0; k < nchains; ++k)
        ACE_HIP(hipMemcpyAsync(latents_out_dev + b0[k] * item_lat, ctxs[k]->xt, Bc[k] * item_lat * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (per_step_ms_host) {
        ACE_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < p->num_steps; ++i) hipEventElapsedTime(&per_step_ms_host[i], evs[i], evs[i + 1]);
    }
    return ACE355_OK;
}

int ace355_dit_set_precision(ace355_dit* h, int precision) {
    ACE_CHECK(h, "set_precision: null handle");
    ACE_CHECK(precision == ACE355_PRECISION_BF16 || precision == ACE355_PRECISION_MXFP8 || precision == ACE355_PRECISION_FP8_WEIGHT_ONLY,
              "set_precision: unknown precision");
    if (!h->finalized) { set_error("set_precision: call ace355_dit_finalize first"); return ACE355_ERR_STATE; }
    ACE_HIP(hipDeviceSynchronize());
    if (h->weights_fp8wo && precision != ACE355_PRECISION_FP8_WEIGHT_ONLY) {
        set_error("set_precision: the weights were rounded through fp8 (fp8_weight_only); load them again before selecting another precision");
        return ACE355_ERR_STATE;
    }
    if (precision == ACE355_PRECISION_FP8_WEIGHT_ONLY) {
        // every nn.Linear of the decoder (the reference's filter keeps all Linears outside tokenizer / detokenizer, init_service_loader.py:
        // 104-113; proj_in / proj_out are Conv1d / ConvTranspose1d and stay): per-output-channel e4m3 + fp32 scale, dequantised to bf16 -
        // applied once, in place, to the packed weights (row packing commutes with a per-row quantiser); the kernels are the bf16 ones
        if (!h->weights_fp8wo) {
            ACE_CHECK(h->mx_allocs.empty(), "set_precision: fp8_weight_only after mxfp8 would round already-copied weights; use a fresh handle");
            const int D = h->D, F = h->F, QD = h->QD, KVD = h->KVD;
            int rc = 0;
            for (LayerW& L : h->layers) {
                if (!rc) rc = launch_fp8_weight_roundtrip(L.wqkv, D, QD + 2 * KVD, D, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(L.wo, QD, D, QD, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(L.wq_c, D, QD, D, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(L.wkv_c, D, 2 * KVD, D, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(L.wo_c, QD, D, QD, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(L.wgu, D, 2 * F, D, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(L.wdown, F, D, F, nullptr);
            }
            for (int e = 0; e < 2 && !rc; ++e) {
                rc = launch_fp8_weight_roundtrip(h->te[e].l1, 256, D, 256, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(h->te[e].l2, D, D, D, nullptr);
                if (!rc) rc = launch_fp8_weight_roundtrip(h->te[e].tp, D, 6 * D, D, nullptr);
            }
            if (!rc) rc = launch_fp8_weight_roundtrip(h->w_cond, D, D, D, nullptr);
            if (rc) return rc;
            ACE_HIP(hipDeviceSynchronize());
            h->weights_fp8wo = true;
            h->nf.key.clear();        // embedding / bias tables are functions of the weights
            if (h->dual.ctx) h->dual.ctx->nf.key.clear();
            for (CondSlot& c : h->slots) c.valid = false;   // cross K/V were projected with the unrounded weights
        }
        h->precision = ACE355_PRECISION_BF16;   // (compute path: the bf16 kernels)
        h->ws_epoch++;
        return ACE355_OK;
    }
    if (precision == ACE355_PRECISION_MXFP8 && h->mx_allocs.empty()) {
        // block-quantise the four big packed projections of every layer once (rows keep their packed order: head-pair q / k rows,
        // [32 gate | 32 up] interleave - quantisation is per row, so the packing commutes with it)
        auto make = [&](const bf16_t* w, int N, int K, MxW* out) -> int {
            if (K % 128 != 0 || N % 256 != 0) return 0;  // tiny configurations: this projection stays bf16
            out->pad = mx_rows_pad(N);
            ALLOC(h->mx_allocs, out->q, (size_t)N * K + 256);
            ALLOC(h->mx_allocs, out->sc, (size_t)(K / 128) * out->pad);
            ACE_HIP(hipMemset(out->sc, 0, (size_t)(K / 128) * out->pad * sizeof(uint32_t)));
            return launch_mx_quant(w, K, N, K, out->q, out->sc, out->pad, nullptr);
        };
        for (LayerW& L : h->layers) {
            L.mx_qkv.bit = 1; L.mx_o.bit = 2; L.mx_gu.bit = 4; L.mx_down.bit = 8; L.mx_qc.bit = 16; L.mx_oc.bit = 32;
            int rc = make(L.wqkv, h->QD + 2 * h->KVD, h->D, &L.mx_qkv);
            if (!rc) rc = make(L.wo, h->D, h->QD, &L.mx_o);
            if (!rc) rc = make(L.wgu, 2 * h->F, h->D, &L.mx_gu);
            if (!rc) rc = make(L.wdown, h->D, h->F, &L.mx_down);
            if (!rc) rc = make(L.wq_c, h->QD, h->D, &L.mx_qc);   // cross-attention q / o projections: used only with ACE355_MX_CROSS=1 (measured: -2 % time, +17 % error)
            if (!rc) rc = make(L.wo_c, h->D, h->QD, &L.mx_oc);
            if (rc) return rc;
        }
        ACE_HIP(hipDeviceSynchronize());
    }
    h->precision = precision;
    h->ws_epoch++;  // a captured sampler graph holds the other precision's launches
    return ACE355_OK;
}

int ace355_dit_poll_errors(ace355_dit* h, void* stream) {
    ACE_CHECK(h, "poll_errors: null handle");
    if (int rc = gemm_splitk_poll(h->sk_cnt, (hipStream_t)stream)) return rc;
    if (int rc = gemm_splitk_poll(h->fk.sk_cnt, (hipStream_t)stream)) return rc;   // (the side stream's launches were joined into `stream` before it could be idle)
    if (h->dual.ctx) return gemm_splitk_poll(h->dual.ctx->sk_cnt, (hipStream_t)stream);
    return ACE355_OK;
}

int ace355_dit_trim_slots(ace355_dit* h, int first_unused) {
    ACE_CHECK(h && first_unused >= 0 && first_unused <= ACE355_MAX_SLOTS, "trim_slots: slot out of range");
    bool any = false;
    for (int i = first_unused; i < ACE355_MAX_SLOTS; ++i) any = any || h->slots[i].kv || h->slots[i].vt || h->slots[i].cross_const;
    if (!any) return ACE355_OK;
    ACE_HIP(hipDeviceSynchronize());   // queued forwards may still read them
    for (int i = first_unused; i < ACE355_MAX_SLOTS; ++i) {
        CondSlot& c = h->slots[i];
        if (c.kv) hipFree(c.kv);
        if (c.vt) hipFree(c.vt);
        if (c.cross_const) hipFree(c.cross_const);
        c = CondSlot{};
    }
    h->cond_epoch++;
    return ACE355_OK;
}

int ace355_dit_set_norm_fold(ace355_dit* h, int enable) {
    ACE_CHECK(h, "set_norm_fold: null handle");
    h->nf.enabled = enable;   // 0 off, 1 default (big-M calls), 2 every call the kernels support (tests)
    h->ws_epoch++;  // a captured sampler graph holds the other variant's launches
    return ACE355_OK;
}

int ace355_dit_set_dual(ace355_dit* h, int mode) {
    ACE_CHECK(h && mode >= 0 && mode <= 2, "set_dual: mode must be 0, 1 or 2");
    h->dual.mode = mode;
    h->ws_epoch++;  // a captured sampler graph holds the other variant's launches
    return ACE355_OK;
}

int ace355_dit_dual_count(ace355_dit* h, int64_t* calls) {
    ACE_CHECK(h && calls, "dual_count: null argument");
    *calls = h->dual.calls;
    return ACE355_OK;
}

int ace355_dit_set_dedup(ace355_dit* h, int enable) {
    ACE_CHECK(h, "set_dedup: null handle");
    h->dedup_on = enable != 0;
    if (h->dual.ctx) h->dual.ctx->dedup_on = h->dedup_on;
    h->ws_epoch++;  // a captured sampler graph holds the other variant's launches
    return ACE355_OK;
}

int ace355_dit_dedup_count(ace355_dit* h, int64_t* forwards) {
    ACE_CHECK(h && forwards, "dedup_count: null argument");
    *forwards = h->dedup_forwards + (h->dual.ctx ? h->dual.ctx->dedup_forwards : 0);
    return ACE355_OK;
}

int ace355_dit_set_graph(ace355_dit* h, int enable) {
    ACE_CHECK(h, "set_graph: null handle");
    h->graph_mode = enable != 0;
    if (!h->graph_mode && h->graph_exec) {
        hipDeviceSynchronize();
        hipGraphExecDestroy(h->graph_exec);
        h->graph_exec = nullptr;
        h->graph_key.clear();
    }
    return ACE355_OK;
}

int ace355_dit_graph_stats(ace355_dit* h, int64_t* captures, int64_t* replays) {
    ACE_CHECK(h, "graph_stats: null handle");
    if (captures) *captures = h->graph_captures;
    if (replays) *replays = h->graph_replays;
    return ACE355_OK;
}

int ace355_dit_set_tap(ace355_dit* h, int layer, float* dst_dev) {
    ACE_CHECK(h && layer >= 0 && layer < h->NL, "set_tap: layer out of range");
    h->tap_dst[layer] = dst_dev;
    return ACE355_OK;
}

int ace355_dit_set_profile(ace355_dit* h, int enable) {
    ACE_CHECK(h, "set_profile: null handle");
    hipDeviceSynchronize();
    for (ace355_dit* c : {h, h->dual.ctx}) {
        if (!c) continue;
        for (auto& e : c->gemm_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
        for (auto& e : c->attn_ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
        c->gemm_ev.clear();
        c->attn_ev.clear();
        c->gemm_flops = c->attn_flops = 0;
        c->gemm_launches = 0;
        c->profile = enable != 0;
    }
    return ACE355_OK;
}

int ace355_dit_get_profile(ace355_dit* h, double* gemm_ms, double* gemm_flops, double* attn_ms, double* attn_flops,
                           int64_t* gemm_launches) {
    ACE_CHECK(h, "get_profile: null handle");
    ACE_HIP(hipDeviceSynchronize());
    // Two chains: every launch of both is timed on its own stream and the two run side by side.  The time reported is the LONGER chain's
    // sum of launch durations (advisor r4: an uneven split - 3 songs as 2 + 1, 5 as 3 + 2 - leaves the longer chain alone at the end, and
    // sum / chains under-reported the chip time there); the flops are both chains', so flops / ms prices the call against the whole chip.
    double g = 0, a = 0, gf = 0, af = 0;
    long gl = 0;
    float ms;
    for (ace355_dit* c : {h, h->dual.ctx}) {
        if (!c || (c != h && c->gemm_ev.empty())) continue;
        double gc = 0, ac = 0;
        for (auto& e : c->gemm_ev) { if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) gc += ms; }
        for (auto& e : c->attn_ev) { if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) ac += ms; }
        g = std::max(g, gc); a = std::max(a, ac);
        gf += c->gemm_flops; af += c->attn_flops; gl += c->gemm_launches;
    }
    if (gemm_ms) *gemm_ms = g;
    if (gemm_flops) *gemm_flops = gf;
    if (attn_ms) *attn_ms = a;
    if (attn_flops) *attn_flops = af;
    if (gemm_launches) *gemm_launches = gl;
    return ACE355_OK;
}

}  // extern "C"
