// cond.hip - the encoder-stack handles behind ace355.h: the condition encoder and the audio-token
// detokenizer of the LM-hint path (row N2, AudioTokenDetokenizer, base.py:862-994, at the end of this file).  N1: weight ingestion/packing and
// AceStepConditionEncoder.forward (base.py:1509-1554) = text projector + AceStepLyricEncoder (base.py:577-731) +
// AceStepTimbreEncoder (base.py:997-1178) + two pack_sequences (base.py:138-169).  Host-side orchestration only: every
// contraction / norm / attention runs on the DiT's kernels (gemm.hip, attn.hip, elementwise.hip).
//
// Differences from the DiT layer that matter here: plain residuals (no gates), un-modulated RMSNorm, and REAL key-padding
// masks on the lyric encoder.  Padding rows are not dead data: the DiT ignores encoder_attention_mask (base.py:1384-1385),
// so the encoder outputs at padded positions flow into the DiT's cross-attention and must match the reference too -
// including query rows with no valid key on sliding layers (AttnArgs::vmean).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <set>
#include <string>
#include <vector>

#include "../../include/ace355.h"
#include "common.h"

using namespace ace355;

namespace {

struct EncLayerW {
    bf16_t *wqkv, *wo, *wgu, *wdown;
    float *n_in, *n_post, *qn, *kn;
};
struct EncoderW {
    int n_layers = 0, in_dim = 0;
    bf16_t* w_embed = nullptr;
    float *b_embed = nullptr, *norm = nullptr;
    std::vector<EncLayerW> layers;
};

}  // namespace

// state shared by every encoder-stack handle: dimensions, weight bag, workspace
struct EncBase {
    int D = 0, F = 0, QD = 0, KVD = 0, HQ = 0, KVH = 0, sliding_window = 0;
    unsigned long long sliding_layer_mask = 0;
    float eps = 1e-6f, theta = 1e6f;
    std::set<std::string> loaded;
    size_t expected_tensors = 0;
    bool finalized = false;
    std::vector<void*> allocs;
    void* stage = nullptr;
    size_t stage_bytes = 0;

    // workspace (sized for the largest M = rows of one encoder call seen so far)
    long ws_rows = 0, ws_in = 0, ws_vt = 0;
    int ws_seqs = 0;
    std::vector<void*> ws_allocs;
    bf16_t *in_bf = nullptr, *xn = nullptr, *qkv = nullptr, *ao = nullptr, *act = nullptr, *vt = nullptr, *vmean = nullptr;
    bf16_t *lyric_out = nullptr, *timbre_out = nullptr, *text_out = nullptr;
    float *h = nullptr, *emb = nullptr;
    int *kvlen_dev = nullptr, *rowsrc_dev = nullptr;
    long rowsrc_cap = 0;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    int rope_S = 0;
    int* sk_cnt = nullptr;   // split-K counters (GemmEpilogue::sk_cnt), allocated with the rope tables
};

struct ace355_cond : EncBase {
    ace355_cond_config cfg;
    EncoderW lyric, timbre;
    bf16_t* w_text = nullptr;
};

struct ace355_detok : EncBase {
    ace355_detok_config cfg;
    EncoderW enc;            // embed_tokens / layers / norm
    float* special = nullptr;  // [pool][D]
    bf16_t* w_out = nullptr;   // [out_dim][D]
    float* b_out = nullptr;
};

// AceStepAudioTokenizer up to the quantizer (base.py:1181-1223): audio_acoustic_proj + AttentionPooler (base.py:734-859)
struct ace355_tok : EncBase {
    ace355_tok_config cfg;
    EncoderW enc;              // attention_pooler.embed_tokens / layers / norm
    float* special = nullptr;  // attention_pooler.special_token [D]
    bf16_t* w_in = nullptr;    // audio_acoustic_proj.weight [D][in_pad] (K padded to a multiple of 64)
    float* b_in = nullptr;
    int in_pad = 64;
    bf16_t* proj = nullptr;    // workspace: projected frames bf16 [R][D]
    long proj_rows = 0;
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
#define ALLOC(bag, ptr, n)                     \
    do {                                       \
        int _rc = dev_alloc(bag, &(ptr), (n)); \
        if (_rc) return _rc;                   \
    } while (0)

struct Dest {
    void* dst;
    int is_bf16, mode;
    long rows, cols, dst_ld, dst_row0;
    int p0;
    bool ignore;
};

// Map a key of AceStepConditionEncoder.state_dict() to its packed destination.
bool resolve(ace355_cond* h, const std::string& name, Dest* d) {
    const long D = h->D, F = h->F, QD = h->QD, KVD = h->KVD;
    auto rows = [&](void* dst, int bf, long r, long c, long ld, long row0) {
        *d = Dest{dst, bf, PACK_ROWS, r, c, ld, row0, 0, false};
        return true;
    };
    if (name == "text_projector.weight") return rows(h->w_text, 1, D, h->cfg.text_hidden_dim, h->cfg.text_hidden_dim, 0);
    for (int e = 0; e < 2; ++e) {
        const std::string p = e == 0 ? "lyric_encoder." : "timbre_encoder.";
        if (name.rfind(p, 0) != 0) continue;
        EncoderW& E = e == 0 ? h->lyric : h->timbre;
        const std::string r = name.substr(p.size());
        if (r == "embed_tokens.weight") return rows(E.w_embed, 1, D, E.in_dim, E.in_dim, 0);
        if (r == "embed_tokens.bias") return rows(E.b_embed, 0, 1, D, D, 0);
        if (r == "norm.weight") return rows(E.norm, 0, 1, D, D, 0);
        if (r == "special_token") {  // parameter of the reference module that its forward never reads (base.py:1015)
            *d = Dest{nullptr, 0, PACK_ROWS, 1, D, D, 0, 0, true};
            return true;
        }
        if (r.rfind("layers.", 0) != 0) return false;
        const size_t dot = r.find('.', 7);
        if (dot == std::string::npos) return false;
        const int li = atoi(r.substr(7, dot - 7).c_str());
        if (li < 0 || li >= E.n_layers) return false;
        EncLayerW& L = E.layers[li];
        const std::string q = r.substr(dot + 1);
        if (q == "input_layernorm.weight") return rows(L.n_in, 0, 1, D, D, 0);
        if (q == "post_attention_layernorm.weight") return rows(L.n_post, 0, 1, D, D, 0);
        // q / k rows in head-pair order for the QKV GEMM's head epilogue (common.h PackMode, gemm.hip mode 4)
        if (q == "self_attn.q_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, QD, D, D, 0, 0, false}; return true; }
        if (q == "self_attn.k_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, KVD, D, D, QD, 0, false}; return true; }
        if (q == "self_attn.v_proj.weight") return rows(L.wqkv, 1, KVD, D, D, QD + KVD);
        if (q == "self_attn.o_proj.weight") return rows(L.wo, 1, D, QD, QD, 0);
        if (q == "self_attn.q_norm.weight") return rows(L.qn, 0, 1, 128, 128, 0);
        if (q == "self_attn.k_norm.weight") return rows(L.kn, 0, 1, 128, 128, 0);
        if (q == "mlp.gate_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 0, false}; return true; }
        if (q == "mlp.up_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 1, false}; return true; }
        if (q == "mlp.down_proj.weight") return rows(L.wdown, 1, D, F, F, 0);
        return false;
    }
    return false;
}

int ensure_rope(EncBase* h, int S, hipStream_t s) {
    if (S <= h->rope_S) return 0;
    int cap = 512;
    while (cap < S) cap *= 2;
    ALLOC(h->allocs, h->rope_cos, (size_t)cap * 64);
    ALLOC(h->allocs, h->rope_sin, (size_t)cap * 64);
    int rc = launch_rope_table(h->rope_cos, h->rope_sin, cap, h->theta, s);
    if (rc) return rc;
    h->rope_S = cap;
    if (!h->sk_cnt) {
        ALLOC(h->allocs, h->sk_cnt, SK_CNT_INTS);
        ACE_HIP(hipMemsetAsync(h->sk_cnt, 0, SK_CNT_INTS * sizeof(int), s));
        if (int prc = gemm_verify_splitk_placement()) return prc;
    }
    return 0;
}

// vt_elems: V^T buffer = seqs x KVD x (S rounded up to 64)
int ensure_workspace(EncBase* h, long rows, long in_elems, int seqs, long vt_elems, hipStream_t s) {
    if (rows <= h->ws_rows && in_elems <= h->ws_in && seqs <= h->ws_seqs && vt_elems <= h->ws_vt) return 0;
    ACE_HIP(hipStreamSynchronize(s));
    for (void* p : h->ws_allocs) hipFree(p);
    h->ws_allocs.clear();
    const long R = std::max(rows, h->ws_rows), I = std::max(in_elems, h->ws_in), V = std::max(vt_elems, h->ws_vt);
    const int Q = std::max(seqs, h->ws_seqs);
    const long D = h->D, QKV = h->QD + 2 * h->KVD;
    ALLOC(h->ws_allocs, h->in_bf, (size_t)I);
    ALLOC(h->ws_allocs, h->h, (size_t)R * D);
    ALLOC(h->ws_allocs, h->xn, (size_t)R * D);
    ALLOC(h->ws_allocs, h->qkv, (size_t)(R + 64) * QKV);  // +64 rows: the last K tile's DMA rows are clamped, not skipped
    ALLOC(h->ws_allocs, h->ao, (size_t)R * h->QD);
    ALLOC(h->ws_allocs, h->act, (size_t)R * h->F);
    ALLOC(h->ws_allocs, h->vt, (size_t)V);
    ALLOC(h->ws_allocs, h->emb, (size_t)R * D);
    ALLOC(h->ws_allocs, h->vmean, (size_t)Q * h->KVD);
    ALLOC(h->ws_allocs, h->kvlen_dev, (size_t)Q);
    ALLOC(h->ws_allocs, h->lyric_out, (size_t)R * D);
    ALLOC(h->ws_allocs, h->timbre_out, (size_t)R * D);
    ALLOC(h->ws_allocs, h->text_out, (size_t)R * D);
    h->ws_rows = R;
    h->ws_in = I;
    h->ws_seqs = Q;
    h->ws_vt = V;
    return 0;
}

// n_layers x AceStepEncoderLayer + final RMSNorm over N sequences of S tokens, on the fp32 stream h->h ([N*S, D], already
// embedded).  kv_len_dev: per-sequence valid key count or null; out: [N*S, D] bf16.
int encoder_layers(EncBase* h, const EncoderW& E, int N, int S, const int* kv_len_dev, bf16_t* out, hipStream_t s) {
    const int M = N * S, D = h->D, F = h->F, QD = h->QD, KVD = h->KVD, QKV = QD + 2 * KVD;
    const int Sp = ((S + 63) / 64) * 64;
    const float eps = h->eps;
    const float scale = 1.0f / sqrtf(128.0f);
    int rc = ensure_rope(h, S, s);
    if (rc) return rc;
    GemmEpilogue ep{};
    for (int li = 0; li < E.n_layers; ++li) {
        const EncLayerW& W = E.layers[li];
        const bool sliding = (h->sliding_layer_mask >> li) & 1ull;
        // self attention (base.py:414-427)
        rc = launch_rmsnorm_mod(h->h, W.n_in, h->xn, M, D, eps, nullptr, nullptr, nullptr, nullptr, 0, S, s);
        if (rc) return rc;
        ep = GemmEpilogue{4, nullptr, nullptr, nullptr, 0, S};  // q / k head-norm + RoPE in the epilogue, as in the DiT
        ep.hn_wq = W.qn, ep.hn_wk = W.kn, ep.hn_cos = h->rope_cos, ep.hn_sin = h->rope_sin;
        ep.hn_q_cols = QD, ep.hn_qk_cols = QD + KVD, ep.hn_eps = eps;
        rc = launch_gemm(h->xn, D, W.wqkv, D, h->qkv, QKV, M, QKV, D, ep, s);
        if (rc) return rc;
        rc = launch_transpose_v(h->qkv, QKV, QD + KVD, N, S, h->KVH, h->vt, Sp, s);
        if (rc) return rc;
        if (kv_len_dev) {
            rc = launch_vmean(h->qkv, QKV, QD + KVD, N, S, h->KVH, h->vmean, s);
            if (rc) return rc;
        }
        AttnArgs a{};
        a.q = h->qkv; a.q_seq_stride = (long)S * QKV; a.q_row_stride = QKV;
        a.k = h->qkv + QD; a.k_seq_stride = (long)S * QKV; a.k_head_stride = 128; a.k_row_stride = QKV;
        a.vt = h->vt; a.vt_seq_stride = (long)h->KVH * 128 * Sp; a.vt_head_stride = 128L * Sp; a.vt_ld = Sp;
        a.use_tab = 0;
        a.out = h->ao; a.o_seq_stride = (long)S * QD; a.o_row_stride = QD;
        a.N = N; a.Sq = S; a.Skv = S; a.Hq = h->HQ; a.Hkv = h->KVH;
        a.window = sliding ? h->sliding_window : -1;
        a.scale = scale;
        a.kv_len = kv_len_dev;
        a.vmean = kv_len_dev ? h->vmean : nullptr;
        rc = launch_attention(a, s);
        if (rc) return rc;
        ep = GemmEpilogue{2, nullptr, nullptr, nullptr, 0, S};  // h += o_proj(attn)
        ep.sk_cnt = h->sk_cnt;
        rc = launch_gemm(h->ao, QD, W.wo, QD, h->h, D, M, D, QD, ep, s);
        if (rc) return rc;
        // SwiGLU MLP (base.py:430-433)
        rc = launch_rmsnorm_mod(h->h, W.n_post, h->xn, M, D, eps, nullptr, nullptr, nullptr, nullptr, 0, S, s);
        if (rc) return rc;
        ep = GemmEpilogue{3, nullptr, nullptr, nullptr, 0, 0};
        rc = launch_gemm(h->xn, D, W.wgu, D, h->act, F, M, 2 * F, D, ep, s);
        if (rc) return rc;
        ep = GemmEpilogue{2, nullptr, nullptr, nullptr, 0, S};
        ep.sk_cnt = h->sk_cnt;
        rc = launch_gemm(h->act, F, W.wdown, F, h->h, D, M, D, F, ep, s);
        if (rc) return rc;
    }
    return launch_rmsnorm_mod(h->h, E.norm, out, M, D, eps, nullptr, nullptr, nullptr, nullptr, 0, S, s);
}

// embed (Linear with bias, base.py:623 / :1070) then the layers.  in_bf: [N*S, in_dim] bf16.
int encoder_stack(EncBase* h, const EncoderW& E, const bf16_t* in_bf, int N, int S, const int* kv_len_dev, bf16_t* out, hipStream_t s) {
    GemmEpilogue ep{1, E.b_embed, nullptr, nullptr, 0, 0};
    int rc = launch_gemm(in_bf, E.in_dim, E.w_embed, E.in_dim, h->h, h->D, N * S, h->D, E.in_dim, ep, s);
    if (rc) return rc;
    return encoder_layers(h, E, N, S, kv_len_dev, out, s);
}

int alloc_encoder(EncBase* h, EncoderW& E, int n_layers, int in_dim) {
    const size_t D = h->D, F = h->F, QD = h->QD, KVD = h->KVD;
    E.n_layers = n_layers;
    E.in_dim = in_dim;
    ALLOC(h->allocs, E.w_embed, D * in_dim);
    ALLOC(h->allocs, E.b_embed, D);
    ALLOC(h->allocs, E.norm, D);
    E.layers.resize(n_layers);
    for (EncLayerW& L : E.layers) {
        ALLOC(h->allocs, L.wqkv, (QD + 2 * KVD) * D);
        ALLOC(h->allocs, L.wo, D * QD);
        ALLOC(h->allocs, L.wgu, 2 * F * D);
        ALLOC(h->allocs, L.wdown, D * F);
        ALLOC(h->allocs, L.n_in, D);
        ALLOC(h->allocs, L.n_post, D);
        ALLOC(h->allocs, L.qn, 128);
        ALLOC(h->allocs, L.kn, 128);
    }
    return 0;
}

}  // namespace

extern "C" {

int ace355_cond_create(const ace355_cond_config* cfg, ace355_cond** out) {
    ACE_CHECK(cfg && out, "cond_create: null argument");
    ACE_CHECK(cfg->head_dim == 128, "cond_create: head_dim must be 128");
    ACE_CHECK(cfg->hidden_size % 256 == 0 && cfg->intermediate_size % 64 == 0, "cond_create: hidden/intermediate size");
    ACE_CHECK(cfg->text_hidden_dim % 64 == 0 && cfg->timbre_hidden_dim % 64 == 0, "cond_create: input dims must be multiples of 64");
    ACE_CHECK(cfg->num_heads % cfg->num_kv_heads == 0, "cond_create: heads % kv_heads");
    ACE_CHECK(cfg->num_lyric_layers >= 0 && cfg->num_lyric_layers <= 64 && cfg->num_timbre_layers >= 0 && cfg->num_timbre_layers <= 64,
              "cond_create: layer counts");
    ace355_cond* h = new ace355_cond();
    h->cfg = *cfg;
    h->D = cfg->hidden_size; h->F = cfg->intermediate_size; h->HQ = cfg->num_heads; h->KVH = cfg->num_kv_heads;
    h->QD = cfg->num_heads * 128; h->KVD = cfg->num_kv_heads * 128;
    h->sliding_window = cfg->sliding_window; h->sliding_layer_mask = cfg->sliding_layer_mask;
    h->eps = cfg->rms_norm_eps; h->theta = cfg->rope_theta;
    int rc = dev_alloc(h->allocs, &h->w_text, (size_t)h->D * cfg->text_hidden_dim);
    if (!rc) rc = alloc_encoder(h, h->lyric, cfg->num_lyric_layers, cfg->text_hidden_dim);
    if (!rc) rc = alloc_encoder(h, h->timbre, cfg->num_timbre_layers, cfg->timbre_hidden_dim);
    if (rc) { ace355_cond_destroy(h); return rc; }
    h->expected_tensors = 1 + (3 + (size_t)cfg->num_lyric_layers * 11) + (3 + (size_t)cfg->num_timbre_layers * 11);
    *out = h;
    return ACE355_OK;
}

void ace355_cond_destroy(ace355_cond* h) {
    if (!h) return;
    hipDeviceSynchronize();
    for (void* p : h->allocs) hipFree(p);
    for (void* p : h->ws_allocs) hipFree(p);
    if (h->stage) hipFree(h->stage);
    if (h->rowsrc_dev) hipFree(h->rowsrc_dev);
    delete h;
}

int ace355_cond_load_tensor(ace355_cond* h, const char* name, const void* data, int dtype, int64_t numel, int is_device) {
    ACE_CHECK(h && name && data, "cond_load_tensor: null argument");
    ACE_CHECK(dtype == ACE355_DTYPE_F32 || dtype == ACE355_DTYPE_BF16, "cond_load_tensor: dtype");
    Dest d;
    if (!resolve(h, name, &d)) {
        set_error(std::string("cond_load_tensor: unknown tensor name '") + name + "'");
        return ACE355_ERR_INVALID;
    }
    if (numel != d.rows * d.cols) {
        set_error(std::string("cond_load_tensor: wrong element count for '") + name + "': got " + std::to_string(numel) +
                  ", expected " + std::to_string(d.rows * d.cols));
        return ACE355_ERR_INVALID;
    }
    if (d.ignore) return ACE355_OK;
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
    int rc = launch_pack(src, dtype, d.dst, d.is_bf16, d.mode, d.rows, d.cols, d.dst_ld, d.dst_row0, d.p0, 0, nullptr);
    if (rc) return rc;
    ACE_HIP(hipDeviceSynchronize());
    h->loaded.insert(name);
    h->finalized = false;
    return ACE355_OK;
}

int ace355_cond_finalize(ace355_cond* h) {
    ACE_CHECK(h, "cond_finalize: null handle");
    if (h->loaded.size() != h->expected_tensors) {
        set_error("cond_finalize: " + std::to_string(h->loaded.size()) + " of " + std::to_string(h->expected_tensors) + " tensors loaded");
        return ACE355_ERR_STATE;
    }
    if (h->stage) { hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
    h->finalized = true;
    return ACE355_OK;
}

int ace355_cond_out_len(int Ll, int Lt, const int32_t* refer_item, int Nref, int B) {
    if (Ll < 0 || Lt < 0 || Nref < 0 || B <= 0 || (Nref > 0 && !refer_item)) return -1;
    std::vector<int> cnt(B, 0);
    int mx = 0;
    for (int i = 0; i < Nref; ++i) {
        if (refer_item[i] < 0 || refer_item[i] >= B) return -1;
        mx = std::max(mx, ++cnt[refer_item[i]]);
    }
    return Ll + mx + Lt;
}

int ace355_cond_encode(ace355_cond* h, const float* text_dev, const int32_t* text_len, int Lt, const float* lyric_dev,
                       const int32_t* lyric_len, int Ll, const float* refer_dev, const int32_t* refer_item, int Nref, int Tref, int B,
                       float* enc_out_dev, int32_t* enc_len_out, void* stream) {
    ACE_CHECK(h && text_dev && text_len && lyric_dev && lyric_len && refer_dev && refer_item && enc_out_dev && enc_len_out,
              "cond_encode: null argument");
    if (!h->finalized) { set_error("cond_encode: call ace355_cond_finalize first"); return ACE355_ERR_STATE; }
    ACE_CHECK(B > 0 && B <= 64 && Lt > 0 && Ll > 0 && Nref > 0 && Nref <= 256 && Tref > 0, "cond_encode: sizes");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->D, TD = h->cfg.text_hidden_dim, AD = h->cfg.timbre_hidden_dim;
    for (int b = 0; b < B; ++b) {
        ACE_CHECK(text_len[b] >= 0 && text_len[b] <= Lt && lyric_len[b] >= 0 && lyric_len[b] <= Ll, "cond_encode: lengths out of range");
    }
    std::vector<int> cnt(B, 0), slot(Nref, 0);
    int max_cnt = 0;
    for (int i = 0; i < Nref; ++i) {
        ACE_CHECK(refer_item[i] >= 0 && refer_item[i] < B, "cond_encode: refer_item out of range");
        slot[i] = cnt[refer_item[i]]++;  // order of appearance inside its batch item (unpack_timbre_embeddings, base.py:1019-1061)
        max_cnt = std::max(max_cnt, cnt[refer_item[i]]);
    }
    const int Lout = Ll + max_cnt + Lt;
    const long rows = std::max<long>({(long)B * Ll, (long)Nref * Tref, (long)B * Lt});
    const long in_elems = std::max<long>({(long)B * Ll * TD, (long)Nref * Tref * AD, (long)B * Lt * TD});
    const long vt_elems = std::max<long>((long)B * (((Ll + 63) / 64) * 64), (long)Nref * (((Tref + 63) / 64) * 64)) * h->KVD;
    int rc = ensure_workspace(h, rows, in_elems, std::max(B, Nref), vt_elems, s);
    if (rc) return rc;

    // text: Linear(text_dim -> D, no bias) (base.py:1541)
    rc = launch_f32_to_bf16(text_dev, h->in_bf, (long)B * Lt * TD, s);
    if (rc) return rc;
    GemmEpilogue ep{0, nullptr, nullptr, nullptr, 0, 0};
    rc = launch_gemm(h->in_bf, TD, h->w_text, TD, h->text_out, D, B * Lt, D, TD, ep, s);
    if (rc) return rc;

    // lyric encoder with its key-padding mask (base.py:1543-1547)
    rc = launch_f32_to_bf16(lyric_dev, h->in_bf, (long)B * Ll * TD, s);
    if (rc) return rc;
    ACE_HIP(hipMemcpyAsync(h->kvlen_dev, lyric_len, sizeof(int) * B, hipMemcpyHostToDevice, s));
    ACE_HIP(hipStreamSynchronize(s));  // lyric_len is caller memory
    rc = encoder_stack(h, h->lyric, h->in_bf, B, Ll, h->kvlen_dev, h->lyric_out, s);
    if (rc) return rc;

    // timbre encoder: no padding mask, token 0 of every reference clip (base.py:1549, :1175)
    rc = launch_f32_to_bf16(refer_dev, h->in_bf, (long)Nref * Tref * AD, s);
    if (rc) return rc;
    rc = encoder_stack(h, h->timbre, h->in_bf, Nref, Tref, nullptr, h->timbre_out, s);
    if (rc) return rc;

    // pack_sequences twice (base.py:1553-1554) with prefix masks = a segmented row gather per batch item:
    //   [lyric valid | timbre valid | text valid | lyric padding | timbre zero rows | text padding]
    // The three sources live in one virtual row space: lyric rows [0, B*Ll), timbre rows (token 0 of clip i) B*Ll + i,
    // text rows B*Ll + Nref + r; they are gathered from three buffers, so build three index tables and gather three times.
    const long total = (long)B * Lout;
    if (total * 3 > h->rowsrc_cap) {
        ACE_HIP(hipStreamSynchronize(s));
        if (h->rowsrc_dev) hipFree(h->rowsrc_dev);
        ACE_HIP(hipMalloc((void**)&h->rowsrc_dev, sizeof(int) * total * 3 + 256));
        h->rowsrc_cap = total * 3;
    }
    std::vector<int> src_l(total, -2), src_t(total, -2), src_x(total, -2);  // -2: row not owned by this source
    std::vector<std::vector<int>> clips(B);
    for (int i = 0; i < Nref; ++i) clips[refer_item[i]].push_back(i);
    for (int b = 0; b < B; ++b) {
        long o = (long)b * Lout;
        const int ll = lyric_len[b], tl = text_len[b], c = cnt[b];
        for (int j = 0; j < ll; ++j) src_l[o++] = b * Ll + j;
        for (int j = 0; j < c; ++j) src_t[o++] = clips[b][j] * Tref;  // token 0 of the clip
        for (int j = 0; j < tl; ++j) src_x[o++] = b * Lt + j;
        for (int j = ll; j < Ll; ++j) src_l[o++] = b * Ll + j;
        for (int j = c; j < max_cnt; ++j) src_t[o++] = -1;              // zero rows of the one-hot unpack
        for (int j = tl; j < Lt; ++j) src_x[o++] = b * Lt + j;
        enc_len_out[b] = ll + c + tl;
    }
    // compact each table to (dst row, src row) pairs handled by one gather over the owned rows only
    std::vector<int> table(total * 3);
    const std::vector<int>* tabs[3] = {&src_l, &src_t, &src_x};
    const bf16_t* bufs[3] = {h->lyric_out, h->timbre_out, h->text_out};
    (void)slot;
    for (int t = 0; t < 3; ++t) memcpy(table.data() + total * t, tabs[t]->data(), sizeof(int) * total);
    ACE_HIP(hipMemcpyAsync(h->rowsrc_dev, table.data(), sizeof(int) * total * 3, hipMemcpyHostToDevice, s));
    for (int t = 0; t < 3; ++t) {
        rc = launch_gather_rows_bf16_f32(bufs[t], D, h->rowsrc_dev + total * t, enc_out_dev, D, total, D, s);
        if (rc) return rc;
    }
    ACE_HIP(hipStreamSynchronize(s));  // `table` is host memory
    return ACE355_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ detokenizer (N2)
namespace {

bool resolve_detok(ace355_detok* h, const std::string& name, Dest* d) {
    const long D = h->D, F = h->F, QD = h->QD, KVD = h->KVD;
    auto rows = [&](void* dst, int bf, long r, long c, long ld, long row0) {
        *d = Dest{dst, bf, PACK_ROWS, r, c, ld, row0, 0, false};
        return true;
    };
    EncoderW& E = h->enc;
    if (name == "embed_tokens.weight") return rows(E.w_embed, 1, D, D, D, 0);
    if (name == "embed_tokens.bias") return rows(E.b_embed, 0, 1, D, D, 0);
    if (name == "norm.weight") return rows(E.norm, 0, 1, D, D, 0);
    if (name == "special_tokens") return rows(h->special, 0, h->cfg.pool_window_size, D, D, 0);
    if (name == "proj_out.weight") return rows(h->w_out, 1, h->cfg.out_dim, D, D, 0);
    if (name == "proj_out.bias") return rows(h->b_out, 0, 1, h->cfg.out_dim, h->cfg.out_dim, 0);
    if (name.rfind("layers.", 0) != 0) return false;
    const size_t dot = name.find('.', 7);
    if (dot == std::string::npos) return false;
    const int li = atoi(name.substr(7, dot - 7).c_str());
    if (li < 0 || li >= E.n_layers) return false;
    EncLayerW& L = E.layers[li];
    const std::string q = name.substr(dot + 1);
    if (q == "input_layernorm.weight") return rows(L.n_in, 0, 1, D, D, 0);
    if (q == "post_attention_layernorm.weight") return rows(L.n_post, 0, 1, D, D, 0);
    // q / k rows in head-pair order for the QKV GEMM's head epilogue (common.h PackMode, gemm.hip mode 4)
    if (q == "self_attn.q_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, QD, D, D, 0, 0, false}; return true; }
    if (q == "self_attn.k_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, KVD, D, D, QD, 0, false}; return true; }
    if (q == "self_attn.v_proj.weight") return rows(L.wqkv, 1, KVD, D, D, QD + KVD);
    if (q == "self_attn.o_proj.weight") return rows(L.wo, 1, D, QD, QD, 0);
    if (q == "self_attn.q_norm.weight") return rows(L.qn, 0, 1, 128, 128, 0);
    if (q == "self_attn.k_norm.weight") return rows(L.kn, 0, 1, 128, 128, 0);
    if (q == "mlp.gate_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 0, false}; return true; }
    if (q == "mlp.up_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 1, false}; return true; }
    if (q == "mlp.down_proj.weight") return rows(L.wdown, 1, D, F, F, 0);
    return false;
}

// shared body of the two load_tensor entry points
int load_packed(EncBase* h, const Dest& d, const char* name, const void* data, int dtype, int64_t numel, int is_device, const char* who) {
    if (numel != d.rows * d.cols) {
        set_error(std::string(who) + ": wrong element count for '" + name + "': got " + std::to_string(numel) + ", expected " +
                  std::to_string(d.rows * d.cols));
        return ACE355_ERR_INVALID;
    }
    if (d.ignore) return ACE355_OK;
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
    int rc = launch_pack(src, dtype, d.dst, d.is_bf16, d.mode, d.rows, d.cols, d.dst_ld, d.dst_row0, d.p0, 0, nullptr);
    if (rc) return rc;
    ACE_HIP(hipDeviceSynchronize());
    h->loaded.insert(name);
    h->finalized = false;
    return ACE355_OK;
}

}  // namespace

extern "C" {

int ace355_detok_create(const ace355_detok_config* cfg, ace355_detok** out) {
    ACE_CHECK(cfg && out, "detok_create: null argument");
    ACE_CHECK(cfg->head_dim == 128, "detok_create: head_dim must be 128");
    ACE_CHECK(cfg->hidden_size % 256 == 0 && cfg->intermediate_size % 64 == 0, "detok_create: hidden/intermediate size");
    ACE_CHECK(cfg->num_heads % cfg->num_kv_heads == 0, "detok_create: heads % kv_heads");
    ACE_CHECK(cfg->num_layers >= 0 && cfg->num_layers <= 64 && cfg->pool_window_size >= 1 && cfg->pool_window_size <= 64 &&
                  cfg->out_dim >= 1, "detok_create: layer count / pool window / out_dim");
    ace355_detok* h = new ace355_detok();
    h->cfg = *cfg;
    h->D = cfg->hidden_size; h->F = cfg->intermediate_size; h->HQ = cfg->num_heads; h->KVH = cfg->num_kv_heads;
    h->QD = cfg->num_heads * 128; h->KVD = cfg->num_kv_heads * 128;
    h->sliding_window = cfg->sliding_window; h->sliding_layer_mask = cfg->sliding_layer_mask;
    h->eps = cfg->rms_norm_eps; h->theta = cfg->rope_theta;
    int rc = alloc_encoder(h, h->enc, cfg->num_layers, cfg->hidden_size);
    if (!rc) rc = dev_alloc(h->allocs, &h->special, (size_t)cfg->pool_window_size * h->D);
    if (!rc) rc = dev_alloc(h->allocs, &h->w_out, (size_t)cfg->out_dim * h->D);
    if (!rc) rc = dev_alloc(h->allocs, &h->b_out, (size_t)cfg->out_dim);
    if (rc) { ace355_detok_destroy(h); return rc; }
    h->expected_tensors = 6 + (size_t)cfg->num_layers * 11;
    *out = h;
    return ACE355_OK;
}

void ace355_detok_destroy(ace355_detok* h) {
    if (!h) return;
    hipDeviceSynchronize();
    for (void* p : h->allocs) hipFree(p);
    for (void* p : h->ws_allocs) hipFree(p);
    if (h->stage) hipFree(h->stage);
    delete h;
}

int ace355_detok_load_tensor(ace355_detok* h, const char* name, const void* data, int dtype, int64_t numel, int is_device) {
    ACE_CHECK(h && name && data, "detok_load_tensor: null argument");
    ACE_CHECK(dtype == ACE355_DTYPE_F32 || dtype == ACE355_DTYPE_BF16, "detok_load_tensor: dtype");
    Dest d;
    if (!resolve_detok(h, name, &d)) {
        set_error(std::string("detok_load_tensor: unknown tensor name '") + name + "'");
        return ACE355_ERR_INVALID;
    }
    return load_packed(h, d, name, data, dtype, numel, is_device, "detok_load_tensor");
}

int ace355_detok_finalize(ace355_detok* h) {
    ACE_CHECK(h, "detok_finalize: null handle");
    if (h->loaded.size() != h->expected_tensors) {
        set_error("detok_finalize: " + std::to_string(h->loaded.size()) + " of " + std::to_string(h->expected_tensors) + " tensors loaded");
        return ACE355_ERR_STATE;
    }
    if (h->stage) { hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
    h->finalized = true;
    return ACE355_OK;
}

int ace355_detok_run(ace355_detok* h, const float* x_dev, int B, int T5, float* out_dev, void* stream) {
    ACE_CHECK(h && x_dev && out_dev, "detok_run: null argument");
    if (!h->finalized) { set_error("detok_run: call ace355_detok_finalize first"); return ACE355_ERR_STATE; }
    ACE_CHECK(B > 0 && T5 > 0 && (long)B * T5 <= (1L << 20), "detok_run: sizes");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->D, P = h->cfg.pool_window_size, OD = h->cfg.out_dim;
    const long R5 = (long)B * T5, R = R5 * P;  // 5 Hz tokens, 25 Hz rows: R5 sequences of P tokens
    const long vt_elems = R5 * h->KVD * (((P + 63) / 64) * 64);
    int rc = ensure_workspace(h, R, R5 * D, (int)R5, vt_elems, s);
    if (rc) return rc;
    // embed_tokens (base.py:888), then one copy per frame of the window + its special token (:889-894)
    rc = launch_f32_to_bf16(x_dev, h->in_bf, R5 * D, s);
    if (rc) return rc;
    GemmEpilogue ep{1, h->enc.b_embed, nullptr, nullptr, 0, 0};
    rc = launch_gemm(h->in_bf, D, h->enc.w_embed, D, h->emb, D, (int)R5, D, D, ep, s);
    if (rc) return rc;
    rc = launch_expand_add(h->emb, h->special, h->h, R5, P, D, s);
    if (rc) return rc;
    // (b t) sequences of P tokens through the encoder layers + norm (:896-987), proj_out (:989), unfold (:991) is a view
    rc = encoder_layers(h, h->enc, (int)R5, P, nullptr, h->xn, s);
    if (rc) return rc;
    ep = GemmEpilogue{1, h->b_out, nullptr, nullptr, 0, 0};
    return launch_gemm(h->xn, D, h->w_out, D, out_dev, OD, (int)R, OD, D, ep, s);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ audio tokenizer (N2, the other direction)
// AceStepAudioTokenizer.forward up to the quantizer (base.py:1206-1218): x [B, T5, P, 64] -> audio_acoustic_proj (Linear 64 -> D) ->
// AttentionPooler: embed_tokens (Linear D -> D), one special token in FRONT of every window of P frames, the (B T5) sequences of P + 1
// tokens through the encoder layers + norm, token 0 of each sequence is the pooled 5 Hz representation [B, T5, D].  Reached by
// prepare_condition for cover tasks without precomputed hints (base.py:1645).  The ResidualFSQ behind it is a [n, D] x [D, 6]
// product + rounding and stays with the caller (lmhints.py).
namespace {

bool resolve_tok(ace355_tok* h, const std::string& name, Dest* d) {
    const long D = h->D, F = h->F, QD = h->QD, KVD = h->KVD;
    auto rows = [&](void* dst, int bf, long r, long c, long ld, long row0) {
        *d = Dest{dst, bf, PACK_ROWS, r, c, ld, row0, 0, false};
        return true;
    };
    if (name == "audio_acoustic_proj.weight") return rows(h->w_in, 1, D, h->cfg.out_dim, h->in_pad, 0);
    if (name == "audio_acoustic_proj.bias") return rows(h->b_in, 0, 1, D, D, 0);
    const std::string p = "attention_pooler.";
    if (name.rfind(p, 0) != 0) return false;
    const std::string r = name.substr(p.size());
    EncoderW& E = h->enc;
    if (r == "embed_tokens.weight") return rows(E.w_embed, 1, D, D, D, 0);
    if (r == "embed_tokens.bias") return rows(E.b_embed, 0, 1, D, D, 0);
    if (r == "norm.weight") return rows(E.norm, 0, 1, D, D, 0);
    if (r == "special_token") return rows(h->special, 0, 1, D, D, 0);
    if (r.rfind("layers.", 0) != 0) return false;
    const size_t dot = r.find('.', 7);
    if (dot == std::string::npos) return false;
    const int li = atoi(r.substr(7, dot - 7).c_str());
    if (li < 0 || li >= E.n_layers) return false;
    EncLayerW& L = E.layers[li];
    const std::string q = r.substr(dot + 1);
    if (q == "input_layernorm.weight") return rows(L.n_in, 0, 1, D, D, 0);
    if (q == "post_attention_layernorm.weight") return rows(L.n_post, 0, 1, D, D, 0);
    if (q == "self_attn.q_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, QD, D, D, 0, 0, false}; return true; }
    if (q == "self_attn.k_proj.weight") { *d = Dest{L.wqkv, 1, PACK_ROWS_HEADPAIR, KVD, D, D, QD, 0, false}; return true; }
    if (q == "self_attn.v_proj.weight") return rows(L.wqkv, 1, KVD, D, D, QD + KVD);
    if (q == "self_attn.o_proj.weight") return rows(L.wo, 1, D, QD, QD, 0);
    if (q == "self_attn.q_norm.weight") return rows(L.qn, 0, 1, 128, 128, 0);
    if (q == "self_attn.k_norm.weight") return rows(L.kn, 0, 1, 128, 128, 0);
    if (q == "mlp.gate_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 0, false}; return true; }
    if (q == "mlp.up_proj.weight") { *d = Dest{L.wgu, 1, PACK_ROWS_IL32, F, D, D, 0, 1, false}; return true; }
    if (q == "mlp.down_proj.weight") return rows(L.wdown, 1, D, F, F, 0);
    return false;
}

}  // namespace

extern "C" {

int ace355_tok_create(const ace355_tok_config* cfg, ace355_tok** out) {
    ACE_CHECK(cfg && out, "tok_create: null argument");
    ACE_CHECK(cfg->head_dim == 128, "tok_create: head_dim must be 128");
    ACE_CHECK(cfg->hidden_size % 256 == 0 && cfg->intermediate_size % 64 == 0, "tok_create: hidden/intermediate size");
    ACE_CHECK(cfg->num_heads % cfg->num_kv_heads == 0, "tok_create: heads % kv_heads");
    ACE_CHECK(cfg->num_layers >= 0 && cfg->num_layers <= 64 && cfg->pool_window_size >= 1 && cfg->pool_window_size <= 63 &&
                  cfg->out_dim >= 1 && cfg->out_dim <= 4096, "tok_create: layer count / pool window / acoustic dim");
    ace355_tok* h = new ace355_tok();
    h->cfg = *cfg;
    h->D = cfg->hidden_size; h->F = cfg->intermediate_size; h->HQ = cfg->num_heads; h->KVH = cfg->num_kv_heads;
    h->QD = cfg->num_heads * 128; h->KVD = cfg->num_kv_heads * 128;
    h->sliding_window = cfg->sliding_window; h->sliding_layer_mask = cfg->sliding_layer_mask;
    h->eps = cfg->rms_norm_eps; h->theta = cfg->rope_theta;
    h->in_pad = ((cfg->out_dim + 63) / 64) * 64;
    int rc = alloc_encoder(h, h->enc, cfg->num_layers, cfg->hidden_size);
    if (!rc) rc = dev_alloc(h->allocs, &h->special, (size_t)h->D);
    if (!rc) rc = dev_alloc(h->allocs, &h->w_in, (size_t)h->D * h->in_pad);
    if (!rc) rc = dev_alloc(h->allocs, &h->b_in, (size_t)h->D);
    if (!rc && hipMemset(h->w_in, 0, (size_t)h->D * h->in_pad * sizeof(bf16_t)) != hipSuccess) rc = ACE355_ERR_HIP;
    if (rc) { ace355_tok_destroy(h); return rc; }
    h->expected_tensors = 6 + (size_t)cfg->num_layers * 11;
    *out = h;
    return ACE355_OK;
}

void ace355_tok_destroy(ace355_tok* h) {
    if (!h) return;
    hipDeviceSynchronize();
    for (void* p : h->allocs) hipFree(p);
    for (void* p : h->ws_allocs) hipFree(p);
    if (h->proj) hipFree(h->proj);
    if (h->stage) hipFree(h->stage);
    delete h;
}

int ace355_tok_load_tensor(ace355_tok* h, const char* name, const void* data, int dtype, int64_t numel, int is_device) {
    ACE_CHECK(h && name && data, "tok_load_tensor: null argument");
    ACE_CHECK(dtype == ACE355_DTYPE_F32 || dtype == ACE355_DTYPE_BF16, "tok_load_tensor: dtype");
    Dest d;
    if (!resolve_tok(h, name, &d)) {
        set_error(std::string("tok_load_tensor: unknown tensor name '") + name + "'");
        return ACE355_ERR_INVALID;
    }
    return load_packed(h, d, name, data, dtype, numel, is_device, "tok_load_tensor");
}

int ace355_tok_finalize(ace355_tok* h) {
    ACE_CHECK(h, "tok_finalize: null handle");
    if (h->loaded.size() != h->expected_tensors) {
        set_error("tok_finalize: " + std::to_string(h->loaded.size()) + " of " + std::to_string(h->expected_tensors) + " tensors loaded");
        return ACE355_ERR_STATE;
    }
    if (h->stage) { hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
    h->finalized = true;
    return ACE355_OK;
}

int ace355_tok_run(ace355_tok* h, const float* x_dev, int B, int T5, float* out_dev, void* stream) {
    ACE_CHECK(h && x_dev && out_dev, "tok_run: null argument");
    if (!h->finalized) { set_error("tok_run: call ace355_tok_finalize first"); return ACE355_ERR_STATE; }
    ACE_CHECK(B > 0 && T5 > 0 && (long)B * T5 <= (1L << 20), "tok_run: sizes");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->D, P = h->cfg.pool_window_size, A = h->cfg.out_dim, S = P + 1;
    const long R5 = (long)B * T5, R = R5 * P, RS = R5 * S;  // 5 Hz tokens, 25 Hz frames, rows of the (P + 1)-token sequences
    const long vt_elems = R5 * h->KVD * (((S + 63) / 64) * 64);
    int rc = ensure_workspace(h, RS, R * h->in_pad, (int)R5, vt_elems, s);
    if (rc) return rc;
    if (R > h->proj_rows) {
        ACE_HIP(hipStreamSynchronize(s));
        if (h->proj) hipFree(h->proj);
        h->proj = nullptr; h->proj_rows = 0;
        ACE_HIP(hipMalloc((void**)&h->proj, (size_t)R * D * sizeof(bf16_t) + 256));
        h->proj_rows = R;
    }
    // frames -> bf16 rows of in_pad columns (zero beyond the acoustic width), audio_acoustic_proj (base.py:1213), embed_tokens (:768)
    rc = launch_pad_rows_bf16(x_dev, h->in_bf, R, A, h->in_pad, s);
    if (rc) return rc;
    GemmEpilogue ep{0, h->b_in, nullptr, nullptr, 0, 0};
    rc = launch_gemm(h->in_bf, h->in_pad, h->w_in, h->in_pad, h->proj, D, (int)R, D, h->in_pad, ep, s);
    if (rc) return rc;
    ep = GemmEpilogue{1, h->enc.b_embed, nullptr, nullptr, 0, 0};
    rc = launch_gemm(h->proj, D, h->enc.w_embed, D, h->emb, D, (int)R, D, D, ep, s);
    if (rc) return rc;
    // [special | P frames] per window (:769-771), the (b t) sequences through the layers + norm (:773-853), token 0 of each (:856-858)
    rc = launch_prepend_special(h->emb, h->special, h->h, R5, P, D, s);
    if (rc) return rc;
    rc = encoder_layers(h, h->enc, (int)R5, S, nullptr, h->xn, s);
    if (rc) return rc;
    return launch_take_token0(h->xn, out_dev, R5, S, D, s);
}

}  // extern "C"
