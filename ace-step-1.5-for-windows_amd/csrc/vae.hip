// vae.hip - the Oobleck VAE handle behind ace355.h: weight-norm fusion + packing at load time, whole-sequence decode
// (architecture per acestep/models/mlx/vae_model.py:190-230) and - SURVEY.md section 8f row N3 - whole-sequence encode
// (vae_model.py:92-116, 148-187, 285-310) as chains of conv_kernel launches (conv.hip).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/ace355.h"
#include "common.h"

using namespace ace355;

namespace {

struct ConvW {
    bf16_t* w = nullptr;   // [N][taps][Cin]
    float* bias = nullptr; // [N] or null
    int N = 0, taps = 0, Cin = 0;
};
struct SnakeP {
    float* ea = nullptr;  // exp(alpha)
    float* ib = nullptr;  // 1 / (exp(beta) + 1e-9)
};
struct ResUnitW {
    SnakeP s1, s2;
    ConvW c1, c2;
    int dil = 1;
};
struct BlockW {
    SnakeP s1;
    ConvW ct;
    int stride = 1, pad = 0, cin = 0, cout = 0;
    ResUnitW ru[3];
};
struct EncBlockW {
    ResUnitW ru[3];
    SnakeP s1;       // tiled `stride` times (virtual channel c' -> channel c' % cin)
    ConvW cd;        // strided conv as 2 taps over stride*cin virtual channels
    int stride = 1, pad = 0, cin = 0, cout = 0;
};
struct RawT {
    float* p = nullptr;
    long n = 0;
};

// ||v_row||_2 per row (one wave per row)
__global__ void rownorm_kernel(const float* __restrict__ v, long cols, float* __restrict__ out, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    double s = 0;
    for (long c = lane; c < cols; c += 64) {
        const double x = v[(long)row * cols + c];
        s += x * x;
    }
    s = wave_sum_d(s);
    if (lane == 0) out[row] = (float)sqrt(s);
}
// Conv1d weight_v [Cout][Cin][K] (+g, norm) -> w[co][k][ci] bf16
__global__ void pack_conv_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ nrm,
                                 bf16_t* __restrict__ w, int Cout, int Cin, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Cout * Cin * K) return;
    const int k = (int)(i % K);
    const long t = i / K;
    const int ci = (int)(t % Cin), co = (int)(t / Cin);
    float x = v[i];
    if (g) x = g[co] * x / (nrm[co] + 1e-9f);
    w[((long)co * K + k) * Cin + ci] = f2bf(x);
}
// ConvTranspose1d weight_v [Cin][Cout][2s] (+g over dim 0) -> w[r*Cout + co][tap][ci]; tap0 <-> k = r + s (x[i0-1]), tap1 <-> k = r (x[i0])
__global__ void pack_convt_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ nrm,
                                  bf16_t* __restrict__ w, int Cin, int Cout, int s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = 2 * s;
    if (i >= (long)Cin * Cout * K) return;
    const int k = (int)(i % K);
    const long t = i / K;
    const int co = (int)(t % Cout), ci = (int)(t / Cout);
    float x = v[i];
    if (g) x = g[ci] * x / (nrm[ci] + 1e-9f);
    const int r = k % s, tap = k >= s ? 0 : 1;
    w[(((long)r * Cout + co) * 2 + tap) * Cin + ci] = f2bf(x);
}
// strided Conv1d weight_v [Cout][C][2s] (+g, norm) -> w[co][g][s' * C + ci] with k = g*s + s'  (2 taps over s*C virtual channels)
__global__ void pack_conv_strided_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ nrm,
                                         bf16_t* __restrict__ w, int Cout, int C, int s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = 2 * s;
    if (i >= (long)Cout * C * K) return;
    const int k = (int)(i % K);
    const long t = i / K;
    const int ci = (int)(t % C), co = (int)(t / C);
    float x = v[i];
    if (g) x = g[co] * x / (nrm[co] + 1e-9f);
    const int gg = k / s, sp = k % s;
    w[((long)co * 2 + gg) * ((long)s * C) + (long)sp * C + ci] = f2bf(x);
}
// first encoder conv (Cin = audio channels, k = 7) as a 1-tap conv over 64 virtual channels: w'[co][k*A + c] = w[co][c][k]
__global__ void pack_conv_im2col_kernel(const float* __restrict__ v, const float* __restrict__ g, const float* __restrict__ nrm,
                                        bf16_t* __restrict__ w, int Cout, int A, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)Cout * 64) return;
    const int cc = (int)(i % 64), co = (int)(i / 64);
    float x = 0.f;
    if (cc < K * A) {
        const int k = cc / A, c = cc % A;
        x = v[((long)co * A + c) * K + k];
        if (g) x = g[co] * x / (nrm[co] + 1e-9f);
    }
    w[i] = f2bf(x);
}
// audio f32 [B][A][L] -> x'[b][l][k*A + c] = audio[b][c][l + k - K/2] (zero outside), 64 bf16 per row
__global__ void audio_im2col_kernel(const float* __restrict__ audio, bf16_t* __restrict__ out, int A, long L, int K, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cc = (int)(i % 64);
    const long bl = i / 64;
    const long l = bl % L, b = bl / L;
    float x = 0.f;
    if (cc < K * A) {
        const int k = cc / A, c = cc % A;
        const long src = l + k - K / 2;
        if (src >= 0 && src < L) x = audio[(b * A + c) * L + src];
    }
    out[i] = f2bf(x);
}
// OobleckDiagonalGaussianDistribution: h f32 [B][2*Z][T] -> z[b][c][t] = mean + (softplus(scale) + 1e-4) * noise (noise null: mean)
__global__ void gaussian_head_kernel(const float* __restrict__ h, const float* __restrict__ noise, float* __restrict__ z, int Z, long T,
                                     long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long t = i % T;
    const long bc = i / T;
    const long c = bc % Z, b = bc / Z;
    const float mean = h[(b * 2 * Z + c) * T + t];
    float out = mean;
    if (noise) {
        const float sc = h[(b * 2 * Z + Z + c) * T + t];
        const float sp = sc > 20.f ? sc : log1pf(expf(sc));
        out = mean + (sp + 1e-4f) * noise[i];
    }
    z[i] = out;
}
__global__ void snake_prep_kernel(const float* __restrict__ alpha, const float* __restrict__ beta, float* __restrict__ ea,
                                  float* __restrict__ ib, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    ea[i] = expf(alpha[i]);
    ib[i] = 1.0f / (expf(beta[i]) + 1e-9f);
}
__global__ void tile_bias_kernel(const float* __restrict__ b, float* __restrict__ out, int C, int reps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * reps) return;
    out[i] = b[i % C];
}
__global__ void cvt_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = bf2f(in[i]);
}

}  // namespace

struct ace355_vae {
    ace355_vae_config cfg;
    int hop = 1;
    std::map<std::string, RawT> raw;
    std::vector<void*> allocs;
    ConvW conv1, conv2;
    SnakeP s_out;
    std::vector<BlockW> blocks;
    bool finalized = false;
    // encoder (optional: present when encoder.* tensors were loaded)
    bool has_encoder = false;
    ConvW e_conv1, e_conv2;
    SnakeP e_s_out;
    std::vector<EncBlockW> e_blocks;
    bf16_t* ain = nullptr;      // im2col'ed audio [B][L][64]
    size_t ain_elems = 0;
    float* e_head = nullptr;    // conv2 output f32 [B][2Z][T]
    size_t e_head_elems = 0;
    // activations
    bf16_t* buf[3] = {nullptr, nullptr, nullptr};
    size_t buf_elems = 0;
    // decode memory policy (ace355_vae_set_decode_budget): bytes the three ping-pong buffers may take before the decode is split
    // into windows (fewer items first, then overlap-discard windows in time)
    size_t decode_budget = (size_t)96 << 30;
    int decode_overlap = 0;          // latent frames of halo per window side; 0: max(16, receptive field + 2)
    int last_plan[3] = {0, 0, 0};    // (items per window, core frames, overlap) of the last decode
    bf16_t* zin = nullptr;
    size_t zin_elems = 0;
    float* scratch = nullptr;
    // profile
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double flops = 0;
    long launches = 0;
};

namespace {

template <typename T>
int valloc(ace355_vae* h, T** p, size_t n) {
    void* q = nullptr;
    ACE_HIP(hipMalloc(&q, n * sizeof(T) + 256));
    h->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}

const RawT* find(ace355_vae* h, const std::string& k) {
    auto it = h->raw.find(k);
    return it == h->raw.end() ? nullptr : &it->second;
}

int need(ace355_vae* h, const std::string& k, long n, const RawT** out) {
    const RawT* r = find(h, k);
    if (!r) { set_error("vae_finalize: missing tensor '" + k + "'"); return ACE355_ERR_STATE; }
    if (r->n != n) { set_error("vae_finalize: wrong size for '" + k + "': got " + std::to_string(r->n) + ", expected " + std::to_string(n)); return ACE355_ERR_INVALID; }
    *out = r;
    return 0;
}

// Conv1d: base.weight_g [Cout,1,1], base.weight_v [Cout,Cin,K] (or fused base.weight), base.bias [Cout]
int build_conv(ace355_vae* h, const std::string& base, int Cout, int Cin, int K, bool has_bias, ConvW* c) {
    const RawT *v = nullptr, *g = nullptr, *b = nullptr;
    int rc;
    const bool fused = find(h, base + ".weight") != nullptr;
    if (fused) {
        if ((rc = need(h, base + ".weight", (long)Cout * Cin * K, &v))) return rc;
    } else {
        if ((rc = need(h, base + ".weight_v", (long)Cout * Cin * K, &v))) return rc;
        if ((rc = need(h, base + ".weight_g", Cout, &g))) return rc;
    }
    if (has_bias && (rc = need(h, base + ".bias", Cout, &b))) return rc;
    c->N = Cout; c->taps = K; c->Cin = Cin;
    if ((rc = valloc(h, &c->w, (size_t)Cout * Cin * K))) return rc;
    float* nrm = nullptr;
    if (g) {
        if ((rc = valloc(h, &nrm, (size_t)Cout))) return rc;
        hipLaunchKernelGGL(rownorm_kernel, dim3((Cout + 3) / 4), dim3(256), 0, 0, v->p, (long)Cin * K, nrm, Cout);
    }
    const long n = (long)Cout * Cin * K;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v->p, g ? g->p : nullptr, nrm, c->w, Cout, Cin, K);
    c->bias = has_bias ? b->p : nullptr;
    ACE_LAUNCH_CHECK();
    return 0;
}

int build_convt(ace355_vae* h, const std::string& base, int Cin, int Cout, int s, ConvW* c) {
    const RawT *v = nullptr, *g = nullptr, *b = nullptr;
    int rc;
    const int K = 2 * s;
    const bool fused = find(h, base + ".weight") != nullptr;
    if (fused) {
        if ((rc = need(h, base + ".weight", (long)Cin * Cout * K, &v))) return rc;
    } else {
        if ((rc = need(h, base + ".weight_v", (long)Cin * Cout * K, &v))) return rc;
        if ((rc = need(h, base + ".weight_g", Cin, &g))) return rc;
    }
    if ((rc = need(h, base + ".bias", Cout, &b))) return rc;
    c->N = s * Cout; c->taps = 2; c->Cin = Cin;
    if ((rc = valloc(h, &c->w, (size_t)Cin * Cout * K))) return rc;
    float* nrm = nullptr;
    if (g) {
        if ((rc = valloc(h, &nrm, (size_t)Cin))) return rc;
        hipLaunchKernelGGL(rownorm_kernel, dim3((Cin + 3) / 4), dim3(256), 0, 0, v->p, (long)Cout * K, nrm, Cin);
    }
    const long n = (long)Cin * Cout * K;
    hipLaunchKernelGGL(pack_convt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v->p, g ? g->p : nullptr, nrm, c->w, Cin, Cout, s);
    if ((rc = valloc(h, &c->bias, (size_t)s * Cout))) return rc;
    hipLaunchKernelGGL(tile_bias_kernel, dim3((s * Cout + 255) / 256), dim3(256), 0, 0, b->p, c->bias, Cout, s);
    ACE_LAUNCH_CHECK();
    return 0;
}

int build_snake(ace355_vae* h, const std::string& base, int C, SnakeP* sp) {
    const RawT *a = nullptr, *b = nullptr;
    int rc;
    if ((rc = need(h, base + ".alpha", C, &a))) return rc;
    if ((rc = need(h, base + ".beta", C, &b))) return rc;
    if ((rc = valloc(h, &sp->ea, (size_t)C))) return rc;
    if ((rc = valloc(h, &sp->ib, (size_t)C))) return rc;
    hipLaunchKernelGGL(snake_prep_kernel, dim3((C + 255) / 256), dim3(256), 0, 0, a->p, b->p, sp->ea, sp->ib, C);
    ACE_LAUNCH_CHECK();
    return 0;
}

int build_conv_strided(ace355_vae* h, const std::string& base, int Cout, int C, int s, ConvW* c) {
    const RawT *v = nullptr, *g = nullptr, *b = nullptr;
    int rc;
    const int K = 2 * s;
    const bool fused = find(h, base + ".weight") != nullptr;
    if (fused) {
        if ((rc = need(h, base + ".weight", (long)Cout * C * K, &v))) return rc;
    } else {
        if ((rc = need(h, base + ".weight_v", (long)Cout * C * K, &v))) return rc;
        if ((rc = need(h, base + ".weight_g", Cout, &g))) return rc;
    }
    if ((rc = need(h, base + ".bias", Cout, &b))) return rc;
    c->N = Cout; c->taps = 2; c->Cin = s * C;
    if ((rc = valloc(h, &c->w, (size_t)Cout * C * K))) return rc;
    float* nrm = nullptr;
    if (g) {
        if ((rc = valloc(h, &nrm, (size_t)Cout))) return rc;
        hipLaunchKernelGGL(rownorm_kernel, dim3((Cout + 3) / 4), dim3(256), 0, 0, v->p, (long)C * K, nrm, Cout);
    }
    const long n = (long)Cout * C * K;
    hipLaunchKernelGGL(pack_conv_strided_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v->p, g ? g->p : nullptr, nrm, c->w, Cout, C, s);
    c->bias = b->p;
    ACE_LAUNCH_CHECK();
    return 0;
}

int build_conv_im2col(ace355_vae* h, const std::string& base, int Cout, int A, int K, ConvW* c) {
    const RawT *v = nullptr, *g = nullptr, *b = nullptr;
    int rc;
    ACE_CHECK(A * K <= 64, "vae_finalize: audio_channels * 7 must fit 64 virtual channels");
    const bool fused = find(h, base + ".weight") != nullptr;
    if (fused) {
        if ((rc = need(h, base + ".weight", (long)Cout * A * K, &v))) return rc;
    } else {
        if ((rc = need(h, base + ".weight_v", (long)Cout * A * K, &v))) return rc;
        if ((rc = need(h, base + ".weight_g", Cout, &g))) return rc;
    }
    if ((rc = need(h, base + ".bias", Cout, &b))) return rc;
    c->N = Cout; c->taps = 1; c->Cin = 64;
    if ((rc = valloc(h, &c->w, (size_t)Cout * 64))) return rc;
    float* nrm = nullptr;
    if (g) {
        if ((rc = valloc(h, &nrm, (size_t)Cout))) return rc;
        hipLaunchKernelGGL(rownorm_kernel, dim3((Cout + 3) / 4), dim3(256), 0, 0, v->p, (long)A * K, nrm, Cout);
    }
    const long n = (long)Cout * 64;
    hipLaunchKernelGGL(pack_conv_im2col_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, v->p, g ? g->p : nullptr, nrm, c->w, Cout, A, K);
    c->bias = b->p;
    ACE_LAUNCH_CHECK();
    return 0;
}

// Snake parameters repeated `reps` times (the strided conv's virtual channels)
int build_snake_tiled(ace355_vae* h, const std::string& base, int C, int reps, SnakeP* sp) {
    SnakeP one;
    int rc = build_snake(h, base, C, &one);
    if (rc) return rc;
    if ((rc = valloc(h, &sp->ea, (size_t)C * reps))) return rc;
    if ((rc = valloc(h, &sp->ib, (size_t)C * reps))) return rc;
    hipLaunchKernelGGL(tile_bias_kernel, dim3((C * reps + 255) / 256), dim3(256), 0, 0, one.ea, sp->ea, C, reps);
    hipLaunchKernelGGL(tile_bias_kernel, dim3((C * reps + 255) / 256), dim3(256), 0, 0, one.ib, sp->ib, C, reps);
    ACE_LAUNCH_CHECK();
    return 0;
}

int build_res_unit(ace355_vae* h, const std::string& r, int C, int dil, ResUnitW* R) {
    int rc;
    R->dil = dil;
    if ((rc = build_snake(h, r + ".snake1", C, &R->s1))) return rc;
    if ((rc = build_conv(h, r + ".conv1", C, C, 7, true, &R->c1))) return rc;
    if ((rc = build_snake(h, r + ".snake2", C, &R->s2))) return rc;
    return build_conv(h, r + ".conv2", C, C, 1, true, &R->c2);
}

int run_conv(ace355_vae* h, const ConvArgs& a, hipStream_t s) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->profile) {
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        h->flops += 2.0 * a.B * (double)a.M * (a.out_mode == 1 ? a.n_real : a.N) * a.taps * a.Cin;
        if (a.w2) h->flops += 2.0 * a.B * (double)a.M * a.N * a.N;  // fused k = 1 stage
        h->launches++;
    }
    int rc = launch_conv(a, s);
    if (h->profile) {
        hipEventRecord(e1, s);
        h->ev.push_back({e0, e1});
    }
    return rc;
}

}  // namespace

extern "C" {

int ace355_vae_create(const ace355_vae_config* cfg, ace355_vae** out) {
    ACE_CHECK(cfg && out, "vae_create: null argument");
    ACE_CHECK(cfg->num_blocks > 0 && cfg->num_blocks <= ACE355_MAX_BLOCKS, "vae_create: num_blocks");
    ACE_CHECK(cfg->decoder_input_channels % 64 == 0 && cfg->decoder_channels % 64 == 0, "vae_create: channels must be multiples of 64");
    ACE_CHECK(cfg->audio_channels >= 1 && cfg->audio_channels <= 32, "vae_create: audio_channels");
    ace355_vae* h = new ace355_vae();
    h->cfg = *cfg;
    h->hop = 1;
    for (int i = 0; i < cfg->num_blocks; ++i) {
        ACE_CHECK(cfg->upsampling_ratios[i] >= 1 && cfg->channel_multiples[i] >= 1, "vae_create: ratios/multiples");
        h->hop *= cfg->upsampling_ratios[i];
    }
    if (const char* e = getenv("ACE355_VAE_BUDGET_MB")) { const long mb = atol(e); if (mb > 0) h->decode_budget = (size_t)mb << 20; }
    *out = h;
    return ACE355_OK;
}

void ace355_vae_destroy(ace355_vae* h) {
    if (!h) return;
    hipDeviceSynchronize();
    for (auto& kv : h->raw) if (kv.second.p) hipFree(kv.second.p);
    for (void* p : h->allocs) hipFree(p);
    for (int i = 0; i < 3; ++i) if (h->buf[i]) hipFree(h->buf[i]);
    if (h->zin) hipFree(h->zin);
    if (h->ain) hipFree(h->ain);
    if (h->e_head) hipFree(h->e_head);
    if (h->scratch) hipFree(h->scratch);
    for (auto& e : h->ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    delete h;
}

int ace355_vae_load_tensor(ace355_vae* h, const char* name, const void* data, int dtype, int64_t numel, int is_device) {
    ACE_CHECK(h && name && data && numel > 0, "vae_load_tensor: null/empty argument");
    ACE_CHECK(dtype == ACE355_DTYPE_F32 || dtype == ACE355_DTYPE_BF16, "vae_load_tensor: dtype");
    const std::string key(name);
    if (key.rfind("decoder.", 0) != 0 && key.rfind("encoder.", 0) != 0) {
        set_error("vae_load_tensor: unknown tensor name '" + key + "' (decoder.* / encoder.* expected)");
        return ACE355_ERR_INVALID;
    }
    RawT& r = h->raw[key];
    if (r.p) { hipFree(r.p); r.p = nullptr; }
    ACE_HIP(hipMalloc((void**)&r.p, (size_t)numel * 4 + 256));
    r.n = numel;
    if (dtype == ACE355_DTYPE_F32) {
        ACE_HIP(hipMemcpy(r.p, data, (size_t)numel * 4, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    } else {
        bf16_t* tmp = nullptr;
        ACE_HIP(hipMalloc((void**)&tmp, (size_t)numel * 2 + 256));
        ACE_HIP(hipMemcpy(tmp, data, (size_t)numel * 2, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_bf16_f32_kernel, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, 0, tmp, r.p, (long)numel);
        ACE_HIP(hipDeviceSynchronize());
        hipFree(tmp);
    }
    h->finalized = false;
    return ACE355_OK;
}

int ace355_vae_finalize(ace355_vae* h) {
    ACE_CHECK(h, "vae_finalize: null handle");
    const ace355_vae_config& c = h->cfg;
    const int nb = c.num_blocks;
    int cm[ACE355_MAX_BLOCKS + 1];
    cm[0] = 1;
    for (int i = 0; i < nb; ++i) cm[i + 1] = c.channel_multiples[i];
    int rc;
    const int C0 = c.decoder_channels * cm[nb];
    if ((rc = build_conv(h, "decoder.conv1", C0, c.decoder_input_channels, 7, true, &h->conv1))) return rc;
    h->blocks.assign(nb, BlockW());
    for (int i = 0; i < nb; ++i) {
        BlockW& B = h->blocks[i];
        B.cin = c.decoder_channels * cm[nb - i];
        B.cout = c.decoder_channels * cm[nb - i - 1];
        B.stride = c.upsampling_ratios[i];
        B.pad = (B.stride + 1) / 2;
        ACE_CHECK(B.cin % 64 == 0 && B.cout % 64 == 0, "vae_finalize: block channels must be multiples of 64");
        const std::string p = "decoder.block." + std::to_string(i);
        if ((rc = build_snake(h, p + ".snake1", B.cin, &B.s1))) return rc;
        if ((rc = build_convt(h, p + ".conv_t1", B.cin, B.cout, B.stride, &B.ct))) return rc;
        const int dils[3] = {1, 3, 9};
        for (int j = 0; j < 3; ++j) {
            ResUnitW& R = B.ru[j];
            R.dil = dils[j];
            const std::string r = p + ".res_unit" + std::to_string(j + 1);
            if ((rc = build_snake(h, r + ".snake1", B.cout, &R.s1))) return rc;
            if ((rc = build_conv(h, r + ".conv1", B.cout, B.cout, 7, true, &R.c1))) return rc;
            if ((rc = build_snake(h, r + ".snake2", B.cout, &R.s2))) return rc;
            if ((rc = build_conv(h, r + ".conv2", B.cout, B.cout, 1, true, &R.c2))) return rc;
        }
    }
    if ((rc = build_snake(h, "decoder.snake1", c.decoder_channels, &h->s_out))) return rc;
    if ((rc = build_conv(h, "decoder.conv2", c.audio_channels, c.decoder_channels, 7, false, &h->conv2))) return rc;
    // encoder half (optional): vae_model.py:148-187.  encoder_hidden_size = 2 * latent channels (mean | scale)
    h->has_encoder = false;
    for (auto& kv : h->raw) if (kv.first.rfind("encoder.", 0) == 0) { h->has_encoder = true; break; }
    if (h->has_encoder) {
        const int EH = 2 * c.decoder_input_channels;
        ACE_CHECK(EH % 64 == 0, "vae_finalize: encoder width must be a multiple of 64");
        if ((rc = build_conv_im2col(h, "encoder.conv1", EH, c.audio_channels, 7, &h->e_conv1))) return rc;
        h->e_blocks.assign(nb, EncBlockW());
        for (int i = 0; i < nb; ++i) {
            EncBlockW& B = h->e_blocks[i];
            B.cin = EH * cm[i];
            B.cout = EH * cm[i + 1];
            B.stride = c.upsampling_ratios[nb - 1 - i];  // downsampling_ratios = reversed upsampling_ratios
            B.pad = (B.stride + 1) / 2;
            ACE_CHECK(B.cin % 64 == 0 && B.cout % 64 == 0, "vae_finalize: encoder block channels must be multiples of 64");
            const std::string p = "encoder.block." + std::to_string(i);
            const int dils[3] = {1, 3, 9};
            for (int j = 0; j < 3; ++j)
                if ((rc = build_res_unit(h, p + ".res_unit" + std::to_string(j + 1), B.cin, dils[j], &B.ru[j]))) return rc;
            if ((rc = build_snake_tiled(h, p + ".snake1", B.cin, B.stride, &B.s1))) return rc;
            if ((rc = build_conv_strided(h, p + ".conv1", B.cout, B.cin, B.stride, &B.cd))) return rc;
        }
        if ((rc = build_snake(h, "encoder.snake1", EH * cm[nb], &h->e_s_out))) return rc;
        if ((rc = build_conv(h, "encoder.conv2", EH, EH * cm[nb], 3, true, &h->e_conv2))) return rc;
    }
    ACE_HIP(hipDeviceSynchronize());
    // raw weight_v / weight_g copies are no longer needed (biases are still referenced)
    for (auto& kv : h->raw) {
        const std::string& k = kv.first;
        const bool keep = k.size() > 5 && k.compare(k.size() - 5, 5, ".bias") == 0;
        if (!keep && kv.second.p) { hipFree(kv.second.p); kv.second.p = nullptr; }
    }
    if (!h->scratch) ACE_HIP(hipMalloc((void**)&h->scratch, 1024));
    h->finalized = true;
    return ACE355_OK;
}

int ace355_vae_hop(const ace355_vae* h) { return h ? h->hop : 0; }

// Residual unit in place on `state` (scratch `tmp`), shared by decode and encode: x + conv_k1(snake2(conv_k7_dil(snake1(x))))
// ACE355_VAE_EPISNAKE (default 1): the Snake of an activation with ONE reader is applied by its producer's epilogue (ConvArgs::osnake_a):
// snake2 of an unfused residual unit by the k = 7 conv (the k = 1 conv then stages plain rows: it evaluated the Snake once per 128-column
// tile, 2 / 4 / 8 times per element at C = 256 / 512 / 1024), the next block's (or the output conv's) Snake by a decoder block's last unit
// and by conv1 (the 2-tap transposed convs evaluated it s * Cout / 128 = 80 / 24 / 8 / 4 / 2 times per element).  0: every Snake in its
// reader's window staging (A/B runs).  `out_snake`: the reader's Snake when the unit's output has that single reader, else null.
static int episnake_on() {
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("ACE355_VAE_EPISNAKE");
        on = e ? atoi(e) : 1;
    }
    return on;
}
static int run_res_unit(ace355_vae* h, const ResUnitW& R, bf16_t*& state, bf16_t*& tmp, int B, long L, int C, hipStream_t s,
                        const SnakeP* out_snake = nullptr) {
    static int fuse = -1;
    if (fuse < 0) {
        const char* e = getenv("ACE355_CONV_FUSE_RU");  // 0: two launches per unit everywhere (A/B runs)
        fuse = e ? atoi(e) : 1;
    }
    if (fuse && C == 128) {
        // C = 128: one workgroup tile spans all channels, so the k = 1 conv runs on the k = 7 result inside the same kernel
        // (conv.hip, ConvArgs::w2).  The unit's output lands in `tmp` (neighbouring tiles still read `state`'s halo rows):
        // the two buffers trade places.
        ConvArgs f{};
        f.x = state; f.x_batch_stride = L * C; f.L_in = (int)L; f.Cin = C;
        f.w = R.c1.w; f.bias = R.c1.bias; f.alpha = R.s1.ea; f.beta = R.s1.ib;
        f.w2 = R.c2.w; f.bias2 = R.c2.bias; f.alpha2 = R.s2.ea; f.beta2 = R.s2.ib;
        f.res = state; f.res_batch_stride = L * C;
        f.y = tmp; f.y_batch_stride = L * C;
        f.B = B; f.M = (int)L; f.N = C; f.taps = 7; f.dil = R.dil; f.center = 3;
        f.y_shift = 0; f.y_valid = L * C; f.out_mode = 0;
        if (out_snake) f.osnake_a = out_snake->ea, f.osnake_b = out_snake->ib;
        const int rcf = run_conv(h, f, s);
        if (rcf) return rcf;
        std::swap(state, tmp);
        return 0;
    }
    ConvArgs a{};
    a.x = state; a.x_batch_stride = L * C; a.L_in = (int)L; a.Cin = C;
    a.w = R.c1.w; a.bias = R.c1.bias; a.alpha = R.s1.ea; a.beta = R.s1.ib;
    a.y = tmp; a.y_batch_stride = L * C;
    a.B = B; a.M = (int)L; a.N = C; a.taps = 7; a.dil = R.dil; a.center = 3;
    a.y_shift = 0; a.y_valid = L * C; a.out_mode = 0;
    const bool epi = episnake_on() != 0;
    if (epi) a.osnake_a = R.s2.ea, a.osnake_b = R.s2.ib;   // tmp = snake2(conv_k7(...)): its only reader is the k = 1 conv below
    int rc = run_conv(h, a, s);
    if (rc) return rc;
    a = ConvArgs{};
    a.x = tmp; a.x_batch_stride = L * C; a.L_in = (int)L; a.Cin = C;
    a.w = R.c2.w; a.bias = R.c2.bias;
    if (!epi) a.alpha = R.s2.ea, a.beta = R.s2.ib;
    if (out_snake) a.osnake_a = out_snake->ea, a.osnake_b = out_snake->ib;
    a.res = state; a.res_batch_stride = L * C;
    a.y = state; a.y_batch_stride = L * C;
    a.B = B; a.M = (int)L; a.N = C; a.taps = 1; a.dil = 1; a.center = 0;
    a.y_shift = 0; a.y_valid = L * C; a.out_mode = 0;
    return run_conv(h, a, s);
}

// Bytes of the three ping-pong activation buffers for a decode of `nb` items x `frames` latent frames (largest stage: L * C).
static size_t decode_elems(const ace355_vae* h, long frames) {
    size_t need = (size_t)frames * h->conv1.N;
    long L = frames;
    for (const BlockW& Bk : h->blocks) {
        L = (L - 1) * Bk.stride - 2 * Bk.pad + 2 * Bk.stride;
        need = std::max(need, (size_t)L * Bk.cout);
    }
    return need;
}
static size_t decode_bytes(const ace355_vae* h, int nb, long frames) { return 3 * (decode_elems(h, frames) * (size_t)nb * 2 + 256); }

// Receptive half-width of one output sample in LATENT frames: conv1 (k7) + per block {2-tap polyphase transposed conv at the input
// rate, three k7 residual units with dilations 1 / 3 / 9 at the output rate} + the k7 output conv.  Everything further away from a
// window edge than this is bit-identical to the whole-sequence decode (same per-element MFMA accumulation order).
static int decode_receptive_frames(const ace355_vae* h) {
    double rf = 3.0, rate = 1.0;
    for (const BlockW& Bk : h->blocks) {
        rf += 1.0 / rate;
        rate *= Bk.stride;
        rf += 3.0 * (1 + 3 + 9) / rate;
    }
    rf += 3.0 / rate;
    return (int)ceil(rf);
}

// One decode window: `nb` items, latent frames [ws, we) of every item (zin rows; rows outside the window are treated as zero,
// which is the true padding at the sequence ends and discarded halo elsewhere); the samples of frames [cs, ce) are written to wav.
static int decode_window(ace355_vae* h, int b0, int nb, int T, int ws, int we, int cs, int ce, float* wav_out_dev, hipStream_t s) {
    const ace355_vae_config& c = h->cfg;
    const int Tw = we - ws;
    bf16_t *cur = h->buf[0], *nxt = h->buf[1], *tmp = h->buf[2];
    long L = Tw;
    int rc;
    ConvArgs a{};
    // conv1: k7, no snake (vae_model.py:224)
    a.x = h->zin + ((size_t)b0 * T + ws) * c.decoder_input_channels; a.x_batch_stride = (long)T * c.decoder_input_channels; a.L_in = Tw;
    a.Cin = c.decoder_input_channels;
    a.w = h->conv1.w; a.bias = h->conv1.bias;
    a.y = cur; a.y_batch_stride = (long)Tw * h->conv1.N;
    a.B = nb; a.M = Tw; a.N = h->conv1.N; a.taps = 7; a.dil = 1; a.center = 3;
    a.y_shift = 0; a.y_valid = (long)Tw * h->conv1.N; a.out_mode = 0;
    const bool epi = episnake_on() != 0 && !h->blocks.empty();
    if (epi) a.osnake_a = h->blocks[0].s1.ea, a.osnake_b = h->blocks[0].s1.ib;   // conv1's only reader: block 0's Snake -> transposed conv
    if ((rc = run_conv(h, a, s))) return rc;

    for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
        const BlockW& Bk = h->blocks[bi];
        // the Snake of whoever reads this block's output (the next block's transposed conv, or the output conv): applied by the last unit
        const SnakeP* next_snake = !epi ? nullptr : (bi + 1 < h->blocks.size() ? &h->blocks[bi + 1].s1 : &h->s_out);
        const long Lout = (L - 1) * Bk.stride - 2 * Bk.pad + 2 * Bk.stride;
        // snake -> ConvTranspose1d as the 2-tap polyphase GEMM (vae_model.py:136-137)
        a = ConvArgs{};
        a.x = cur; a.x_batch_stride = L * Bk.cin; a.L_in = (int)L; a.Cin = Bk.cin;
        a.w = Bk.ct.w; a.bias = Bk.ct.bias;
        if (!epi) a.alpha = Bk.s1.ea, a.beta = Bk.s1.ib;   // (epi: `cur` already holds snake(x), written by its producer)
        a.y = nxt; a.y_batch_stride = Lout * Bk.cout;
        a.B = nb; a.M = (int)L + 1; a.N = Bk.ct.N; a.taps = 2; a.dil = 1; a.center = 1;
        a.y_shift = -(long)Bk.pad * Bk.cout; a.y_valid = Lout * Bk.cout; a.out_mode = 0;
        if ((rc = run_conv(h, a, s))) return rc;
        for (int j = 0; j < 3; ++j)  // x + conv_k1(snake2(conv_k7_dil(snake1(x)))), in place on the block state (vae_model.py:79-87)
            if ((rc = run_res_unit(h, Bk.ru[j], nxt, tmp, nb, Lout, Bk.cout, s, j == 2 ? next_snake : nullptr))) return rc;
        std::swap(cur, nxt);
        L = Lout;
    }
    // snake -> conv k7 -> [B, audio, hop * T] fp32 (vae_model.py:228-229): the window's core samples go straight to their place
    const long hop = h->hop, Lall = hop * T;
    a = ConvArgs{};
    a.x = cur; a.x_batch_stride = L * c.decoder_channels; a.L_in = (int)L; a.Cin = c.decoder_channels;
    a.w = h->conv2.w; a.bias = nullptr;
    if (!epi) a.alpha = h->s_out.ea, a.beta = h->s_out.ib;
    a.y = wav_out_dev + (size_t)b0 * c.audio_channels * Lall; a.y_batch_stride = Lall * c.audio_channels;
    a.B = nb; a.M = (int)L; a.N = c.audio_channels; a.taps = 7; a.dil = 1; a.center = 3;
    a.out_mode = 1; a.n_real = c.audio_channels;
    a.ncl_ld = Lall; a.ncl_off = (long)cs * hop; a.ncl_m_lo = (int)((cs - ws) * hop); a.ncl_m_hi = (int)((ce - ws) * hop);
    return run_conv(h, a, s);
}

// Window plan of a decode under the activation budget (the reference bounds decode memory by policy: chunk sizes by free VRAM,
// H/memory_utils.py:48-83, overlap-discard windows, H/vae_decode_chunks.py:51-112): whole batch and whole sequence when the three
// ping-pong buffers fit; else fewer items per window (exact: items are independent); else windows of `tc` core frames with `ov`
// halo frames on either side, the halo decoded and discarded.
static int decode_plan(const ace355_vae* h, int B, int T, size_t budget, int* nb_out, int* tc_out, int* ov_out) {
    const int rf = decode_receptive_frames(h);
    int ov = h->decode_overlap > 0 ? h->decode_overlap : std::max(16, ((rf + 2 + 7) / 8) * 8);
    ACE_CHECK(ov >= rf, "vae_decode: the window overlap must cover the decoder's receptive field");
    int nb = B, tc = T;
    if (decode_bytes(h, B, T) > budget) {
        const size_t per_item = decode_bytes(h, 1, T);
        if (per_item <= budget) nb = (int)std::max<size_t>(1, std::min<size_t>(B, budget / per_item));
        else {
            nb = 1;
            // bytes grow linearly in the window length: largest window that fits, minus the two halos
            const size_t per_frame = decode_bytes(h, 1, 1024) / 1024 + 1;
            const long win = (long)(budget / per_frame);
            tc = (int)std::min<long>(T, win - 2L * ov);
            ACE_CHECK(tc >= ov, "vae_decode: the activation budget is smaller than one decode window (raise ACE355_VAE_BUDGET_MB)");
            tc = (tc / 8) * 8;  // (whole 8-frame groups: window starts stay 16-byte aligned in the latent rows)
            // several items per window when they fit
            const size_t per_win = decode_bytes(h, 1, std::min<long>(T, tc + 2L * ov));
            nb = (int)std::max<size_t>(1, std::min<size_t>(B, budget / per_win));
        }
    }
    *nb_out = nb; *tc_out = tc; *ov_out = ov;
    return 0;
}

int ace355_vae_decode(ace355_vae* h, const float* z_dev, int B, int T, float* wav_out_dev, void* stream) {
    RoctxRange r_dec("ace355.vae_decode");
    ACE_CHECK(h && z_dev && wav_out_dev, "vae_decode: null argument");
    if (!h->finalized) { set_error("vae_decode: call ace355_vae_finalize first"); return ACE355_ERR_STATE; }
    ACE_CHECK(B > 0 && T > 0, "vae_decode: empty problem");
    ACE_CHECK((long)T * h->hop < (1L << 31), "vae_decode: sequence too long (2^31 samples per item)");
    hipStream_t s = (hipStream_t)stream;
    const ace355_vae_config& c = h->cfg;
    size_t budget = h->decode_budget;
    int nb = B, tc = T, ov = 0;
    // allocation with the reference's out-of-memory policy (H/vae_decode_chunks.py:40-81 retries smaller instead of failing):
    // a refused hipMalloc halves the budget and re-plans, down to one minimal window
    for (int attempt = 0;; ++attempt) {
        int rc = decode_plan(h, B, T, budget, &nb, &tc, &ov);
        if (rc) return rc;
        const long win = std::min<long>(T, tc == T ? T : tc + 2L * ov);
        const size_t need_elems = decode_elems(h, win) * (size_t)nb;
        if (need_elems <= h->buf_elems) break;
        ACE_HIP(hipStreamSynchronize(s));
        bool ok = true;
        for (int i = 0; i < 3; ++i) {
            if (h->buf[i]) hipFree(h->buf[i]);
            h->buf[i] = nullptr;
        }
        h->buf_elems = 0;
        for (int i = 0; i < 3 && ok; ++i) ok = hipMalloc((void**)&h->buf[i], need_elems * 2 + 256) == hipSuccess;
        if (ok) { h->buf_elems = need_elems; break; }
        (void)hipGetLastError();
        for (int i = 0; i < 3; ++i) { if (h->buf[i]) hipFree(h->buf[i]); h->buf[i] = nullptr; }
        budget = std::min(budget, decode_bytes(h, nb, win)) / 2;
        if (attempt >= 12 || budget < decode_bytes(h, 1, 3L * ov + 8)) {
            set_error("vae_decode: out of device memory even for one minimal decode window");
            return ACE355_ERR_HIP;
        }
    }
    const size_t zel = (size_t)B * T * c.decoder_input_channels;
    if (zel > h->zin_elems) {
        ACE_HIP(hipStreamSynchronize(s));
        if (h->zin) hipFree(h->zin);
        h->zin = nullptr; h->zin_elems = 0;
        ACE_HIP(hipMalloc((void**)&h->zin, zel * 2 + 256));
        h->zin_elems = zel;
    }
    int rc = launch_ncl_to_nlc(z_dev, h->zin, B, c.decoder_input_channels, T, s);
    if (rc) return rc;
    h->last_plan[0] = nb; h->last_plan[1] = tc; h->last_plan[2] = ov;
    for (int b0 = 0; b0 < B; b0 += nb) {
        const int nbw = std::min(nb, B - b0);
        for (int cs = 0; cs < T; cs += tc) {
            const int ce = std::min(T, cs + tc);
            const int ws = tc == T ? 0 : std::max(0, cs - ov), we = tc == T ? T : std::min(T, ce + ov);
            if ((rc = decode_window(h, b0, nbw, T, ws, we, cs, ce, wav_out_dev, s))) return rc;
        }
    }
    return ACE355_OK;
}

int ace355_vae_set_decode_budget(ace355_vae* h, int64_t bytes, int overlap_frames) {
    ACE_CHECK(h, "vae_set_decode_budget: null handle");
    ACE_CHECK(bytes >= 0 && overlap_frames >= 0, "vae_set_decode_budget: negative argument");
    if (bytes > 0) h->decode_budget = (size_t)bytes;
    if (overlap_frames > 0) {
        ACE_CHECK(overlap_frames >= decode_receptive_frames(h) && overlap_frames % 8 == 0,
                  "vae_set_decode_budget: the overlap must cover the receptive field and be a multiple of 8 frames");
        h->decode_overlap = overlap_frames;
    }
    return ACE355_OK;
}

int ace355_vae_decode_plan(ace355_vae* h, int B, int T, int32_t* items_per_window, int32_t* core_frames, int32_t* overlap_frames,
                           int64_t* activation_bytes) {
    ACE_CHECK(h && B > 0 && T > 0, "vae_decode_plan: bad argument");
    if (!h->finalized) { set_error("vae_decode_plan: call ace355_vae_finalize first"); return ACE355_ERR_STATE; }
    int nb, tc, ov;
    int rc = decode_plan(h, B, T, h->decode_budget, &nb, &tc, &ov);
    if (rc) return rc;
    if (items_per_window) *items_per_window = nb;
    if (core_frames) *core_frames = tc;
    if (overlap_frames) *overlap_frames = ov;
    if (activation_bytes) *activation_bytes = (int64_t)decode_bytes(h, nb, std::min<long>(T, tc == T ? T : tc + 2L * ov));
    return ACE355_OK;
}

static int encode_group(ace355_vae* h, const float* audio_dev, const float* noise_dev, int B, int64_t L, float* latents_out_dev, hipStream_t s);

int ace355_vae_encode(ace355_vae* h, const float* audio_dev, const float* noise_dev, int B, int64_t L, float* latents_out_dev,
                      void* stream) {
    ACE_CHECK(h && audio_dev && latents_out_dev, "vae_encode: null argument");
    if (!h->finalized) { set_error("vae_encode: call ace355_vae_finalize first"); return ACE355_ERR_STATE; }
    if (!h->has_encoder) { set_error("vae_encode: no encoder.* tensors were loaded"); return ACE355_ERR_STATE; }
    ACE_CHECK(B > 0 && L > 0 && L < (1L << 30), "vae_encode: sizes");
    // same activation budget as the decode (ace355_vae_set_decode_budget): above it the batch is encoded in groups of items (exact:
    // items are independent; the reference's 30 s / 2 s-overlap encode tiling, H/vae_encode.py:47-86, is the same kind of memory bound)
    const ace355_vae_config& c = h->cfg;
    const int EH = 2 * c.decoder_input_channels, Z = c.decoder_input_channels;
    const size_t per_item = (size_t)L * EH * 2 * 3 + (size_t)L * 64 * 2;
    int nb = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, h->decode_budget / std::max<size_t>(per_item, 1)));
    long T = L;
    for (const EncBlockW& Bk : h->e_blocks) T = (T + 2 * Bk.pad - 2 * Bk.stride) / Bk.stride + 1;
    ACE_CHECK(T > 0, "vae_encode: audio shorter than one latent frame");
    for (int b0 = 0; b0 < B; b0 += nb) {
        const int n = std::min(nb, B - b0);
        int rc = encode_group(h, audio_dev + (size_t)b0 * c.audio_channels * L, noise_dev ? noise_dev + (size_t)b0 * Z * T : nullptr, n, L,
                              latents_out_dev + (size_t)b0 * Z * T, (hipStream_t)stream);
        if (rc) return rc;
    }
    return ACE355_OK;
}

static int encode_group(ace355_vae* h, const float* audio_dev, const float* noise_dev, int B, int64_t L, float* latents_out_dev, hipStream_t s) {
    const ace355_vae_config& c = h->cfg;
    const int EH = 2 * c.decoder_input_channels, Z = c.decoder_input_channels;
    // stage lengths: Conv1d(k = 2s, stride s, pad ceil(s/2)): L_out = floor((L + 2 pad - 2 s) / s) + 1
    long T = L;
    for (const EncBlockW& Bk : h->e_blocks) {
        T = (T + 2 * Bk.pad - 2 * Bk.stride) / Bk.stride + 1;
        ACE_CHECK(T > 0, "vae_encode: audio shorter than one latent frame");
    }
    const size_t need_elems = (size_t)B * L * EH;
    if (need_elems > h->buf_elems) {
        ACE_HIP(hipStreamSynchronize(s));
        for (int i = 0; i < 3; ++i) {
            if (h->buf[i]) hipFree(h->buf[i]);
            h->buf[i] = nullptr;
            ACE_HIP(hipMalloc((void**)&h->buf[i], need_elems * 2 + 256));
        }
        h->buf_elems = need_elems;
    }
    const size_t ael = (size_t)B * L * 64;
    if (ael > h->ain_elems) {
        ACE_HIP(hipStreamSynchronize(s));
        if (h->ain) hipFree(h->ain);
        ACE_HIP(hipMalloc((void**)&h->ain, ael * 2 + 256));
        h->ain_elems = ael;
    }
    const size_t hel = (size_t)B * EH * T;
    if (hel > h->e_head_elems) {
        ACE_HIP(hipStreamSynchronize(s));
        if (h->e_head) hipFree(h->e_head);
        ACE_HIP(hipMalloc((void**)&h->e_head, hel * 4 + 256));
        h->e_head_elems = hel;
    }
    hipLaunchKernelGGL(audio_im2col_kernel, dim3((unsigned)((ael + 255) / 256)), dim3(256), 0, s, audio_dev, h->ain, c.audio_channels, (long)L, 7,
                       (long)ael);
    ACE_LAUNCH_CHECK();

    bf16_t *cur = h->buf[0], *nxt = h->buf[1], *tmp = h->buf[2];
    int rc;
    ConvArgs a{};
    // conv1 (k7 over the audio channels, vae_model.py:166) as one tap over the im2col'ed rows
    a.x = h->ain; a.x_batch_stride = L * 64; a.L_in = (int)L; a.Cin = 64;
    a.w = h->e_conv1.w; a.bias = h->e_conv1.bias;
    a.y = cur; a.y_batch_stride = L * EH;
    a.B = B; a.M = (int)L; a.N = EH; a.taps = 1; a.dil = 1; a.center = 0;
    a.y_shift = 0; a.y_valid = L * EH; a.out_mode = 0;
    if ((rc = run_conv(h, a, s))) return rc;
    long Lc = L;
    const bool epi = episnake_on() != 0;   // (run_res_unit: the Snakes with one reader move to their producers, as in the decoder)
    for (size_t bi = 0; bi < h->e_blocks.size(); ++bi) {
        const EncBlockW& Bk = h->e_blocks[bi];
        for (int j = 0; j < 3; ++j)   // (s1 is tiled `stride` times: its first cin entries are the per-channel parameters)
            if ((rc = run_res_unit(h, Bk.ru[j], cur, tmp, B, Lc, Bk.cin, s, (epi && j == 2) ? &Bk.s1 : nullptr))) return rc;
        // snake -> Conv1d(k = 2s, stride s) as 2 taps over the shifted view x'[r][c'] = x_flat[(r*s - pad)*cin + c'] (vae_model.py:113-115)
        const long Lo = (Lc + 2 * Bk.pad - 2 * Bk.stride) / Bk.stride + 1;
        a = ConvArgs{};
        a.x = cur; a.x_batch_stride = Lc * Bk.cin; a.L_in = (int)Lo + 1; a.Cin = Bk.stride * Bk.cin;
        a.x_shift = -(long)Bk.pad * Bk.cin; a.x_valid = Lc * Bk.cin;
        a.w = Bk.cd.w; a.bias = Bk.cd.bias;
        if (!epi) a.alpha = Bk.s1.ea, a.beta = Bk.s1.ib;
        if (epi && bi + 1 == h->e_blocks.size()) a.osnake_a = h->e_s_out.ea, a.osnake_b = h->e_s_out.ib;   // read by conv2 only
        a.y = nxt; a.y_batch_stride = Lo * Bk.cout;
        a.B = B; a.M = (int)Lo; a.N = Bk.cout; a.taps = 2; a.dil = 1; a.center = 0;
        a.y_shift = 0; a.y_valid = Lo * Bk.cout; a.out_mode = 0;
        if ((rc = run_conv(h, a, s))) return rc;
        std::swap(cur, nxt);
        Lc = Lo;
    }
    // snake -> conv k3 -> f32 [B][2Z][T] (vae_model.py:185-186), then the diagonal Gaussian head (:296-302)
    const int Cl = h->e_conv2.Cin;
    a = ConvArgs{};
    a.x = cur; a.x_batch_stride = Lc * Cl; a.L_in = (int)Lc; a.Cin = Cl;
    a.w = h->e_conv2.w; a.bias = h->e_conv2.bias;
    if (!epi || h->e_blocks.empty()) a.alpha = h->e_s_out.ea, a.beta = h->e_s_out.ib;
    a.y = h->e_head; a.y_batch_stride = (long)EH * Lc;
    a.B = B; a.M = (int)Lc; a.N = EH; a.taps = 3; a.dil = 1; a.center = 1;
    a.out_mode = 1; a.n_real = EH;
    if ((rc = run_conv(h, a, s))) return rc;
    const long total = (long)B * Z * Lc;
    hipLaunchKernelGGL(gaussian_head_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h->e_head, noise_dev, latents_out_dev, Z, Lc, total);
    ACE_LAUNCH_CHECK();
    return ACE355_OK;
}

int ace355_vae_latent_frames(const ace355_vae* h, int64_t L) {
    if (!h || !h->finalized || !h->has_encoder || L <= 0) return -1;
    long T = L;
    for (const EncBlockW& Bk : h->e_blocks) {
        T = (T + 2 * Bk.pad - 2 * Bk.stride) / Bk.stride + 1;
        if (T <= 0) return -1;
    }
    return (int)T;
}

int ace355_vae_set_profile(ace355_vae* h, int enable) {
    ACE_CHECK(h, "vae_set_profile: null handle");
    hipDeviceSynchronize();
    for (auto& e : h->ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    h->ev.clear();
    h->flops = 0;
    h->launches = 0;
    h->profile = enable != 0;
    return ACE355_OK;
}

int ace355_vae_get_profile(ace355_vae* h, double* conv_ms, double* conv_flops, int64_t* conv_launches) {
    ACE_CHECK(h, "vae_get_profile: null handle");
    ACE_HIP(hipDeviceSynchronize());
    double g = 0;
    float ms;
    for (auto& e : h->ev) if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) g += ms;
    if (conv_ms) *conv_ms = g;
    if (conv_flops) *conv_flops = h->flops;
    if (conv_launches) *conv_launches = h->launches;
    return ACE355_OK;
}

}  // extern "C"
