/*
 * ace355.h - C ABI of the MI355X-native (gfx950) ACE-Step 1.5 denoise + decode path.
 *
 * The reference (sdbds/ACE-Step-1.5-for-windows) has no FFI; its backend seam is a pair
 * of Python mixin methods added for MLX.  This header is what a ctypes binding
 * behind that seam talks to (INTEGRATION.md shows the stub).  Every entry point cites the
 * reference interface it replaces (paths relative to /root/reference/acestep/):
 *
 *   base.py = models/base/modeling_acestep_v15_base.py      H/ = core/generation/handler/
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes; no torch / C++ types.
 *   - "dev" pointers are HIP device pointers on the current device; "host" pointers are
 *     ordinary host memory.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - every function returns ACE355_OK (0) or an error code; ace355_last_error() returns a
 *     thread-local message (HIP error string included).  Nothing aborts the process
 *     (reference contract: backend failure = exception caught at the seam,
 *     H/service_generate_execute.py:189-191, H/vae_decode.py:44-48).
 *   - handles are not thread-safe; one generation in flight per handle (SURVEY.md section 5,
 *     "Race detection": the reference serialises requests per handler).
 */
#ifndef ACE355_H
#define ACE355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACE355_OK 0
#define ACE355_ERR_INVALID 1 /* bad argument / shape / unknown tensor name */
#define ACE355_ERR_HIP 2     /* a HIP runtime call failed */
#define ACE355_ERR_STATE 3   /* call order violated (e.g. forward before finalize) */
#define ACE355_ERR_UNSUPPORTED 4

#define ACE355_DTYPE_F32 0
#define ACE355_DTYPE_BF16 1

#define ACE355_MAX_BLOCKS 8
#define ACE355_MAX_SEQS 64
#define ACE355_MAX_SLOTS 32 /* resident condition slots per DiT handle: 8 cover + 8 non-cover per-item conditions + null + spare */

const char* ace355_last_error(void);
int ace355_version(void);
/* Measurement aid (bench.py `box_probe`; no reference counterpart): dense bf16 MFMA rate of THIS board on a pure
 * v_mfma_f32_32x32x16_bf16 loop with random operands, 8 waves per CU, no memory traffic (TFLOP/s; the boxes of the pool differ by a
 * few percent under the same power cap, and a bench line is only comparable across boxes beside this number).  Synchronises the device. */
int ace355_box_probe_mfma(int iters, double* tflops_out);
/* Tuning knob of the bf16 GEMM (no reference counterpart; process-wide, returns the previous mode).  K rotation: in a launch that gives
 * every CU exactly one output tile (the o_proj / down projections at the metric batch), all eight L2s would miss the same K slice of the
 * operands at the same moment; with the rotation on, the workgroups of XCD x start at K step x * nk / 8 and wrap, so a slice's first reader
 * pays the HBM latency and the others find it in the memory-side cache (K step 1.12 -> 1.06 us, - 1 % per 8-song pass).  The fp32 accumulation
 * ORDER of an output element then depends on the XCD region its tile falls in, i.e. on the launch shape: the same request still gives the
 * same bits, but a song's bits depend on the batch it is part of (3e-3 rel L2 apart, both at the same distance from the reference).
 * mode 0: off (one K order whatever the shape; what the bit-identity tests of the CFG fork pin), 1 (default; env ACE355_GEMM_KROT):
 * launches with N <= 2048, 2: every one-round launch.  Not a per-handle setting: call it before the first request.
 * Mode 0 is the library's "launch-shape-independent arithmetic" switch as a whole: the attention launcher then also keeps to its one-walk
 * kernels (the 12-wave key-split kernel of the 96-row launches merges two partial softmaxes per row, and which launches are 96-row ones
 * depends on how many sequences share them), so a song computed alone and inside a larger batch agree bit for bit wherever both runs take
 * the GQA kernels (tests/test_metric_shapes_gpu.py::test_120s_forward_vs_reference_golden_and_batch16). */
int ace355_gemm_set_k_rotation(int mode);

/* ------------------------------------------------------------------------------------------
 * DiT decoder (AceStepDiTModel, base.py:1240-1507) + sampler (generate_audio, base.py:1783-1989)
 * ---------------------------------------------------------------------------------------- */
typedef struct ace355_dit ace355_dit;

/* Fields of AceStepConfig the path uses (models/base/configuration_acestep_v15.py:148-263). */
typedef struct ace355_dit_config {
    int32_t hidden_size;       /* 2048 */
    int32_t intermediate_size; /* 6144 */
    int32_t num_layers;        /* 24 */
    int32_t num_heads;         /* 16 */
    int32_t num_kv_heads;      /* 8 */
    int32_t head_dim;          /* 128 (only 128 is supported) */
    int32_t sliding_window;    /* 128: |i-j| <= window, inclusive (base.py:99-105) */
    int32_t patch_size;        /* 2 (only 2 is supported) */
    int32_t in_channels;       /* 192 = 128 context + 64 latent */
    int32_t out_channels;      /* 64 */
    float rms_norm_eps;        /* 1e-6 */
    float rope_theta;          /* 1e6 */
    uint64_t sliding_layer_mask; /* bit i set = layer i is "sliding_attention" (cfg.py:251-254) */
} ace355_dit_config;

/* Replaces H/mlx_dit_init.py:9-43 (_init_mlx_dit) + models/mlx/dit_convert.py:69-84. */
int ace355_dit_create(const ace355_dit_config* cfg, ace355_dit** out);
void ace355_dit_destroy(ace355_dit* h);

/* Upload one tensor of AceStepDiTModel.state_dict() by its reference name (e.g.
 * "layers.3.self_attn.q_proj.weight", "proj_in.1.weight", "time_embed.linear_1.bias",
 * "scale_shift_table").  `data` holds `numel` elements of `dtype`, on host or device.
 * The library converts / packs into its own MFMA-friendly bf16 layout (fused QKV,
 * interleaved gate|up, patchify as GEMM).  Unknown names or wrong numel -> ERR_INVALID. */
int ace355_dit_load_tensor(ace355_dit* h, const char* name, const void* data, int dtype, int64_t numel, int is_device);
/* Verifies that every tensor was loaded; must be called once before set_condition/forward. */
int ace355_dit_finalize(ace355_dit* h);

/* Encoder conditioning for one "slot" (cond, null, non-cover cond, ...).
 * Runs condition_embedder (base.py:1359) and every layer's cross-attention
 * K = k_norm(k_proj(.)), V = v_proj(.) ONCE (the reference's EncoderDecoderCache,
 * base.py:312-329, 1875) and keeps them resident until the slot is overwritten.
 * enc: dev f32 [rows, hidden]; rows == L, or rows == 1 to broadcast one row over L keys
 * (null_condition_emb.expand_as, base.py:1907).  slot in [0, ACE355_MAX_SLOTS). */
int ace355_dit_set_condition(ace355_dit* h, int slot, const float* enc_dev, int rows, int L, void* stream);

/* One decoder forward = AceStepDiTModel.forward (base.py:1303-1507) with use_cache=True and
 * a filled cross cache.  x dev f32 [N,T,64]; ctx dev f32 [N,T,128]; t, t_r host [N];
 * slots host [N] (which condition slot each sequence attends to); v_out dev f32 [N,T,64]. */
int ace355_dit_forward(ace355_dit* h, const float* x_dev, const float* ctx_dev, const float* t_host,
                       const float* t_r_host, const int32_t* slots_host, int N, int T, float* v_out_dev, void* stream);

/* Knobs of generate_audio (base.py:1796-1812) that survive past prepare_condition. */
typedef struct ace355_sample_params {
    int32_t num_steps;          /* len(t_sched) - 1 */
    const float* t_sched_host;  /* [num_steps+1] fp32, base.py:1864-1867 (caller applies shift / custom timesteps) */
    float guidance_scale;       /* diffusion_guidance_sale; > 1 enables CFG batch doubling (base.py:1905) */
    float cfg_interval_start;   /* base.py:1945 */
    float cfg_interval_end;
    int32_t infer_method;       /* 0 = "ode" (base.py:1974-1979); 1 = "sde" (base.py:1968-1973), needs sde_noise_dev */
    int32_t use_adg;            /* 1 = angle-based guidance (apg_guidance.py:107-180) instead of APG inside the cfg interval */
    int32_t cond_slot;          /* slot of the cover/main condition */
    int32_t null_slot;          /* slot holding null_condition_emb (ignored when guidance <= 1) */
    int32_t cover_switch_step;  /* = int(steps * audio_cover_strength); >= num_steps: never switch (base.py:1916-1927) */
    int32_t non_cover_slot;     /* slot + context used from cover_switch_step on */
    const float* ctx_non_cover_dev; /* dev f32 [B,T,128] or NULL */
    const float* sde_noise_dev;     /* "sde" only: dev f32 [num_steps,B,T,64], the per-step randn_like(x) draws of base.py:1777
                                     * (the reference draws them unseeded on the model device; the caller owns the RNG) */
    /* Per-item conditions inside one call: the reference's loop takes B distinct encoder_hidden_states rows
     * (base.py:1905-1911, switch :1916-1927).  host int32 [B] each, or NULL = every item uses cond_slot / non_cover_slot.
     * All slots of one call must hold the same encoder length L. */
    const int32_t* cond_slots_host;
    const int32_t* non_cover_slots_host;
    int32_t sde_next_from_sched;    /* "sde" renoise level after step i: 0 = 1 - (i+1)/num_steps (base / sft, base.py:1972);
                                     * 1 = t_sched[i+1] (turbo model, models/turbo/modeling_acestep_v15_turbo.py:1980-1984) */
} ace355_sample_params;

/* The sampling loop of generate_audio (base.py:1913-1981): CFG doubling, steps x {decoder forward,
 * APG (apg_guidance.py:33-56, momentum -0.75, norm threshold 2.5, fp64 projection), Euler update}.
 * xt0 dev f32 [B,T,64] is the initial state (noise, or the cover renoise of base.py:1887, prepared by
 * the caller on the host with the CPU generator: SURVEY.md section 7.2 "Noise parity");
 * ctx dev f32 [B,T,128]; latents_out dev f32 [B,T,64] ("target_latents").
 * per_step_ms_host (optional, [num_steps]) receives HIP-event step times. */
int ace355_dit_sample(ace355_dit* h, const float* xt0_dev, const float* ctx_dev, int B, int T,
                      const ace355_sample_params* p, float* latents_out_dev, float* per_step_ms_host, void* stream);

/* Compute precision of the four big projections of every layer (QKV, o_proj, gate|up, down).  BF16 (default) = the reference GPU
 * path's dtype.  MXFP8 = BASELINE configs[4] "fp8 MFMA": OCP MXFP8 operands (e4m3, one E8M0 scale per 32 K elements; weights
 * quantised once here, activations per launch) on v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate, same epilogues; launches with
 * fewer than 1536 token rows keep the bf16 kernels.  The reference's counterpart is torchao fp8 on the DiT Linears
 * (handler/init_service_loader.py:89-113; absent here: parity unpinned, tolerance stated in DESIGN.md).  Call after finalize.
 * FP8_WEIGHT_ONLY = the reference's default fp8 knob (`quantization="fp8_weight_only"`, torchao Float8WeightOnlyConfig,
 * init_service_loader.py:95-97): weights as e4m3 with one fp32 scale per output channel, activations and arithmetic bf16.  gfx950 has
 * no bf16 x fp8 MFMA, so this mode applies the NUMERICS (every Linear weight rounded through per-channel e4m3 once, in place) and
 * runs the bf16 kernels: same speed and memory as BF16.  One way: load the weights again to go back. */
#define ACE355_PRECISION_BF16 0
#define ACE355_PRECISION_MXFP8 1
#define ACE355_PRECISION_FP8_WEIGHT_ONLY 2
int ace355_dit_set_precision(ace355_dit* h, int precision);

/* hipGraph replay of the sampling loop: with enable != 0, ace355_dit_sample captures its launch
 * sequence (steps x ~360 kernels) into a graph on first use and replays it for every later call with the same shapes, schedule,
 * knobs, slot layout and stream; any change re-captures.  Results are identical to the eager path of the same chain configuration (same
 * kernels, same order; by default a captured call runs as ONE sampler chain - two chains become two graph branches that the runtime
 * places on streams of its own choosing, measured slower - while an eager request of 2-3 songs runs as two: ace355_dit_set_dual).
 * Off by default; ACE355_SAMPLE_GRAPH=1 in the environment turns it on at handle creation.  graph_stats: captures / replays so far. */
int ace355_dit_set_graph(ace355_dit* h, int enable);
int ace355_dit_graph_stats(ace355_dit* h, int64_t* captures, int64_t* replays);

/* RMSNorm folding (on by default; ACE355_NORM_FOLD=0 in the environment or enable = 0 here turns it off): in bf16 ace355_dit_sample calls
 * of >= 64 token rows (ACE355_NORM_FOLD_MIN_ROWS; since round 3 it pays at every real size) the three RMSNorms of a decoder layer
 * (base.py:493-533) are not kernels of their own: the residual GEMM that finishes hidden_states also writes bf16(h * g) and the rows' sums of
 * squares, the consuming projection applies rsqrt(mean(h^2) + eps) and the modulation shift's projection (shift W^T, precomputed per step of
 * the schedule at the start of the call) to its fp32 accumulators.  Same math, one bf16 rounding placed differently.  A row's sum of squares
 * is gathered per aligned 128-column group - (b0 + b1) + (b2 + b3) over its four 32-column blocks in fp32, then 2^-24 fixed point and integer adds -
 * so it does not depend on the GEMM tile form a row was computed in (round 6).
 * enable: 0 off, 1 default, 2 every call the kernels support whatever its size (tests). */
int ace355_dit_set_norm_fold(ace355_dit* h, int enable);

/* Dual-chain sampler.  The songs of a request are independent through the whole sampling loop of generate_audio (per-item noise,
 * per-item CFG / APG; base.py:1783-1989), so a call with B >= 2 songs can run as TWO half-batch samplers - songs [0, ceil(B/2)) on the
 * caller's stream, the rest on a side stream that sits on a hardware queue of its own (checked once per caller stream: the runtime
 * shares a few queues between all streams, and two streams on one queue run strictly in turn).  mode 0: one chain; 1 (default,
 * ACE355_DUAL): two chains when one chain's launches would under-fill the chip: <= 2400 token rows in the whole request
 * (ACE355_DUAL_MAX_ROWS: 2-3 songs of 30 s under CFG), or a row count whose tiles leave more than 15 % of the CU slots of their
 * rounds empty (30 s songs: 5, 6, 9, 10 per request; not 4, 7, 8) - there one chain's launches fill the CUs the other leaves idle (per
 * request incl. decode, one chain -> two: 2 x 30 s 199 -> 194 ms, 3: 266 -> 251, 5: 397 -> 358, 6: 430 -> 409, 9: 658 -> 582);
 * 2: two chains whenever B >= 2 (4 songs: 290 -> 304 ms, the metric batch of 8: 487 -> 519 ms: one chain already gives every CU a
 * tile and the two chains only share the power cap).  Each song's
 * result is what a one-chain call with that song's half-batch returns.  Captured like any other launch under ace355_dit_set_graph.
 * dual_count: calls that ran as two chains so far. */
int ace355_dit_set_dual(ace355_dit* h, int mode);
int ace355_dit_dual_count(ace355_dit* h, int64_t* calls);

/* Layer-0 de-duplication under classifier-free guidance (on by default; ace355_dit_set_dedup(h, 0) or ACE355_DEDUP0=0 in the environment switch it off).  The
 * reference feeds the decoder x = cat([xt, xt]) with the same context latents and timestep for both copies (base.py:1905-1911, 1929): up to
 * the first cross-attention (base.py:499-511 of layer 0) the conditional and the null copy of a song are the same numbers.
 * ace355_dit_sample therefore runs layer 0's first norm, QKV projection and self-attention on the conditional half only and lets the
 * o_proj GEMM read that half's attention output for both halves.  dedup_count: forwards that took the shortcut so far - counted when a
 * forward is ENQUEUED: replays of a captured sampler graph (ace355_dit_set_graph) run the shortcut without advancing it. */
int ace355_dit_set_dedup(ace355_dit* h, int enable);
int ace355_dit_dedup_count(ace355_dit* h, int64_t* forwards);

/* Test / debug hook: after decoder layer `layer` (0-based) of every following forward, copy the fp32 residual stream
 * hidden_states [N*S, hidden] (the layer's output, base.py:539) to dst_dev; dst_dev NULL clears the tap.  Lets the parity
 * tests compare the per-layer activations the golden fixtures hold, not only the final velocity. */
int ace355_dit_set_tap(ace355_dit* h, int layer, float* dst_dev);

/* Asynchronous device-side conditions of the calls queued so far on `stream` (synchronises it): today the ordered split-K of the
 * small-M residual GEMMs, whose parts wait for their turn with a bound - a missed turn is counted on the device, reported here as
 * ACE355_ERR_HIP and the turn counters are reset.  ACE355_OK otherwise.  The Python host calls it where it synchronises anyway. */
int ace355_dit_poll_errors(ace355_dit* h, void* stream);
/* Release the cross-K/V buffers of condition slots [first_unused, ACE355_MAX_SLOTS) (a request with many distinct conditions
 * leaves up to 32 x 150 MB resident otherwise).  Synchronises the device when there is something to free. */
int ace355_dit_trim_slots(ace355_dit* h, int first_unused);
/* Work counters for roofline accounting: algorithmic FLOPs of the last forward (SURVEY.md section 8d formula)
 * and GEMM-only HIP-event time when profiling was enabled with ace355_dit_set_profile(h, 1). */
int ace355_dit_set_profile(ace355_dit* h, int enable);
int ace355_dit_get_profile(ace355_dit* h, double* gemm_ms, double* gemm_flops, double* attn_ms, double* attn_flops,
                           int64_t* gemm_launches);

/* ------------------------------------------------------------------------------------------
 * Oobleck VAE decoder (diffusers AutoencoderOobleck.decode; restated in models/mlx/vae_model.py:190-230)
 * ---------------------------------------------------------------------------------------- */
typedef struct ace355_vae ace355_vae;

typedef struct ace355_vae_config {
    int32_t decoder_channels;       /* 128 */
    int32_t decoder_input_channels; /* 64 */
    int32_t audio_channels;         /* 2 */
    int32_t num_blocks;             /* 5 */
    int32_t channel_multiples[ACE355_MAX_BLOCKS]; /* [1,2,4,8,16] */
    int32_t upsampling_ratios[ACE355_MAX_BLOCKS]; /* decoder order, e.g. [10,6,4,4,2] (product = hop 1920) */
} ace355_vae_config;

/* Replaces H/mlx_vae_init.py:12-96 (_init_mlx_vae) + models/mlx/vae_convert.py. */
int ace355_vae_create(const ace355_vae_config* cfg, ace355_vae** out);
void ace355_vae_destroy(ace355_vae* h);
/* Names are AutoencoderOobleck.state_dict() keys of the decoder half:
 * "decoder.conv1.weight_g|weight_v|bias", "decoder.block.{i}.snake1.alpha|beta",
 * "decoder.block.{i}.conv_t1.*", "decoder.block.{i}.res_unit{1,2,3}.{snake1,conv1,snake2,conv2}.*",
 * "decoder.snake1.*", "decoder.conv2.weight_g|weight_v".  Weight-norm is fused at finalize
 * (w = g*v/(||v||+1e-9), vae_convert.py:18-34).  A pre-fused "<conv>.weight" is accepted too. */
int ace355_vae_load_tensor(ace355_vae* h, const char* name, const void* data, int dtype, int64_t numel, int is_device);
int ace355_vae_finalize(ace355_vae* h);
/* Replaces vae.decode(z).sample at H/vae_decode_chunks.py:42,95 / H/generate_music_decode.py:172-177 and
 * _mlx_vae_decode (H/mlx_vae_decode_native.py:31-72).  z dev f32 [B,64,T] (the reference's
 * [B,C,T] layout); wav_out dev f32 [B,2,hop*T].  Whole-sequence decode when the three activation buffers fit the decode
 * budget: equals the reference's overlap-discard tiling away from fp summation order.  Above the
 * budget the call splits itself: fewer items per pass first (exact), then overlap-discard windows in time whose halo covers
 * the decoder's receptive field (bit-identical to the whole-sequence result); a refused allocation halves the budget and
 * re-plans instead of failing (the reference's policy: H/memory_utils.py:48-83, H/vae_decode_chunks.py:40-112). */
int ace355_vae_decode(ace355_vae* h, const float* z_dev, int B, int T, float* wav_out_dev, void* stream);
/* Decode memory policy.  bytes > 0: budget of the three ping-pong activation buffers (default 96 GiB, or ACE355_VAE_BUDGET_MB at
 * create); overlap_frames > 0: halo per window side in latent frames (multiple of 8, >= the receptive field; default
 * max(16, receptive field + 2); the reference uses 64, H/vae_decode.py:16).  Zero leaves a setting unchanged. */
int ace355_vae_set_decode_budget(ace355_vae* h, int64_t bytes, int overlap_frames);
/* The window plan ace355_vae_decode would use for (B, T): items per window, core frames per window (== T: whole sequence),
 * halo frames, bytes of the three activation buffers.  Any output pointer may be NULL. */
int ace355_vae_decode_plan(ace355_vae* h, int B, int T, int32_t* items_per_window, int32_t* core_frames, int32_t* overlap_frames,
                           int64_t* activation_bytes);
int ace355_vae_hop(const ace355_vae* h);
/* Encoder half (vae_model.py:92-116, 148-187, 285-310): available when the encoder.* keys of
 * AutoencoderOobleck.state_dict() were loaded before ace355_vae_finalize.  audio dev f32 [B, audio_channels, L];
 * latents_out dev f32 [B, 64, T] with T = ace355_vae_latent_frames(h, L) (= L / hop for multiples of the hop);
 * noise dev f32 [B, 64, T] -> `latent_dist.sample()` = mean + (softplus(scale) + 1e-4) * noise
 * (handler/vae_encode.py:66); noise NULL -> the mean (`latent_dist.mode()`). */
int ace355_vae_encode(ace355_vae* h, const float* audio_dev, const float* noise_dev, int B, int64_t L, float* latents_out_dev,
                      void* stream);
int ace355_vae_latent_frames(const ace355_vae* h, int64_t L);
int ace355_vae_set_profile(ace355_vae* h, int enable);
int ace355_vae_get_profile(ace355_vae* h, double* conv_ms, double* conv_flops, int64_t* conv_launches);

/* ------------------------------------------------------------------------------------------
 * Condition encoder: AceStepConditionEncoder.forward, base.py:1509-1554
 * = text_projector + AceStepLyricEncoder (base.py:577-731) + AceStepTimbreEncoder (base.py:997-1178)
 * + pack_sequences x2 (base.py:138-169).  Replaces the call `self.encoder(...)` of prepare_condition
 * (base.py:1626-1633); its output is what ace355_dit_set_condition takes.
 * ---------------------------------------------------------------------------------------- */
typedef struct ace355_cond ace355_cond;

typedef struct ace355_cond_config {
    int32_t hidden_size;        /* 2048 */
    int32_t intermediate_size;  /* 6144 */
    int32_t num_heads;          /* 16 */
    int32_t num_kv_heads;       /* 8 */
    int32_t head_dim;           /* 128 (only value supported) */
    int32_t text_hidden_dim;    /* 1024: text and lyric embedding width */
    int32_t timbre_hidden_dim;  /* 64: reference-audio latent width */
    int32_t num_lyric_layers;   /* 8 */
    int32_t num_timbre_layers;  /* 4 */
    int32_t sliding_window;     /* 128 */
    uint64_t sliding_layer_mask; /* bit i set: encoder layer i is "sliding_attention" (config.layer_types[i]) */
    float rms_norm_eps;         /* 1e-6 */
    float rope_theta;           /* 1e6 */
} ace355_cond_config;

int ace355_cond_create(const ace355_cond_config* cfg, ace355_cond** out);
void ace355_cond_destroy(ace355_cond* h);
/* name = a key of AceStepConditionEncoder.state_dict() (`model.encoder` of the reference checkpoint), e.g.
 * "lyric_encoder.layers.3.self_attn.q_proj.weight"; "timbre_encoder.special_token" is accepted and ignored
 * (the reference's forward never reads it).  Same contract as ace355_dit_load_tensor. */
int ace355_cond_load_tensor(ace355_cond* h, const char* name, const void* data, int dtype, int64_t numel, int is_device);
int ace355_cond_finalize(ace355_cond* h);
/* Rows per item of the packed output: Ll + (max number of reference clips of one item) + Lt; < 0 on bad arguments. */
int ace355_cond_out_len(int Ll, int Lt, const int32_t* refer_item_host, int Nref, int B);
/* One encoder forward.  text dev f32 [B,Lt,text_dim], lyric dev f32 [B,Ll,text_dim], refer dev f32
 * [Nref,Tref,timbre_dim] (refer_audio_acoustic_hidden_states_packed); text_len / lyric_len host int32 [B]: the
 * attention masks in prefix form (mask[b][j] = j < len[b]; anything else must take the PyTorch path);
 * refer_item host int32 [Nref] = refer_audio_order_mask.  Writes enc_out dev f32 [B, out_len, hidden]
 * (encoder_hidden_states, valid tokens first exactly as pack_sequences orders them, padding rows included because the
 * DiT attends them: base.py:1384-1385) and enc_len_out host int32 [B] (encoder_attention_mask in prefix form). */
int ace355_cond_encode(ace355_cond* h, const float* text_dev, const int32_t* text_len_host, int Lt, const float* lyric_dev,
                       const int32_t* lyric_len_host, int Ll, const float* refer_dev, const int32_t* refer_item_host,
                       int Nref, int Tref, int B, float* enc_out_dev, int32_t* enc_len_out_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * Audio-token detokenizer of the LM-hint path: AudioTokenDetokenizer.forward,
 * base.py:886-994, as called by _decode_audio_codes_to_latents (handler/audio_codes.py:49-66) and by
 * prepare_condition (base.py:1646-1647).  Input = the quantizer's output for the 5 Hz audio codes, output =
 * lm_hints_25Hz.
 * ---------------------------------------------------------------------------------------- */
typedef struct ace355_detok ace355_detok;

typedef struct ace355_detok_config {
    int32_t hidden_size;        /* 2048 */
    int32_t intermediate_size;  /* 6144 */
    int32_t num_heads;          /* 16 */
    int32_t num_kv_heads;       /* 8 */
    int32_t head_dim;           /* 128 (only value supported) */
    int32_t num_layers;         /* 2 (num_attention_pooler_hidden_layers) */
    int32_t pool_window_size;   /* 5: 25 Hz frames per 5 Hz token */
    int32_t out_dim;            /* 64 (audio_acoustic_hidden_dim) */
    int32_t sliding_window;     /* 128 */
    uint64_t sliding_layer_mask;
    float rms_norm_eps;
    float rope_theta;
} ace355_detok_config;

int ace355_detok_create(const ace355_detok_config* cfg, ace355_detok** out);
void ace355_detok_destroy(ace355_detok* h);
/* name = a key of AudioTokenDetokenizer.state_dict() (`model.detokenizer` of the reference checkpoint). */
int ace355_detok_load_tensor(ace355_detok* h, const char* name, const void* data, int dtype, int64_t numel, int is_device);
int ace355_detok_finalize(ace355_detok* h);
/* x dev f32 [B, T5, hidden] -> out dev f32 [B, T5 * pool_window_size, out_dim]. */
int ace355_detok_run(ace355_detok* h, const float* x_dev, int B, int T5, float* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Audio tokenizer, the opposite direction of the same row: AceStepAudioTokenizer.forward up to its quantizer
 * (base.py:1181-1218) = audio_acoustic_proj (Linear 64 -> hidden) + AttentionPooler (base.py:734-859: embed_tokens, one
 * special token in front of every window of pool_window_size frames, num_layers encoder layers over the (B T5) sequences
 * of pool_window_size + 1 tokens, norm, token 0).  Replaces `self.tokenizer(x)` of model.tokenize (base.py:1580-1591), which
 * prepare_condition reaches for cover tasks without precomputed hints (base.py:1645).  The ResidualFSQ behind the pooler
 * (third-party vector_quantize_pytorch: Linear hidden -> 6, bounded rounding, Linear 6 -> hidden) stays with the caller
 * (ace355/lmhints.py).  Same configuration fields as the detokenizer; `out_dim` is the acoustic width (64).
 * ---------------------------------------------------------------------------------------- */
typedef struct ace355_tok ace355_tok;
typedef ace355_detok_config ace355_tok_config;
int ace355_tok_create(const ace355_tok_config* cfg, ace355_tok** out);
void ace355_tok_destroy(ace355_tok* h);
/* name = a key of AceStepAudioTokenizer.state_dict() (`model.tokenizer`) outside `quantizer.`: "audio_acoustic_proj.*",
 * "attention_pooler.*". */
int ace355_tok_load_tensor(ace355_tok* h, const char* name, const void* data, int dtype, int64_t numel, int is_device);
int ace355_tok_finalize(ace355_tok* h);
/* x dev f32 [B, T5 * pool_window_size, out_dim] (25 Hz frames, whole windows) -> out dev f32 [B, T5, hidden]. */
int ace355_tok_run(ace355_tok* h, const float* x_dev, int B, int T5, float* out_dev, void* stream);

/* The two projections of the residual FSQ quantizer on the LM-hint path (`model.tokenizer.quantizer.project_in / project_out`:
 * hidden -> 6 and 6 -> hidden; reached from H/audio_codes.py:47-66 `_decode_audio_codes_to_latents` through
 * `quantizer.get_output_from_indices`, and from base.py:1206-1218 `AceStepAudioTokenizer.forward`): out f32 [M, N] = x f32 [M, K] W^T
 * (W f32 [N, K], nn.Linear layout) + b (f32 [N] or NULL), accumulated in fp64 - the caller rounds project_in's result to FSQ digits, so a
 * half-way value must not depend on a summation order.  All pointers device memory. */
int ace355_linear_f32(const float* x_dev, const float* w_dev, const float* b_dev, float* out_dev, int64_t M, int N, int K, void* stream);

/* Post-decode peak clip, H/generate_music_decode.py:191-195: per item, if any peak > 1 divide
 * every item by clamp(peak, min=1).  wav dev f32 [B, per_item]. */
int ace355_peak_normalize(float* wav_dev, int B, int64_t per_item, void* stream);
/* NaN/Inf/all-zero latent guard, H/generate_music_decode.py:66-77: flags_host[0]=has_nan_or_inf, [1]=all_zero. */
int ace355_latent_check(const float* lat_dev, int64_t numel, int32_t* flags_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * Output stage: what acestep/inference.py:649-726 does per item after the decode,
 * batched.  Replaces normalize_audio (acestep/audio_utils.py:24-62) and AudioSaver.save_audio / save_batch
 * (audio_utils.py:65-215, 259-313) for the formats whose codecs are containers over PCM: "flac" (PCM_16, the
 * reference default), "wav" / "wav32" (IEEE float32, what torchaudio's soundfile backend writes for a float32
 * tensor).  "mp3" / "opus" / "aac" go through ffmpeg in the reference and stay there.
 * ------------------------------------------------------------------------------------------ */
#define ACE355_AUDIO_FLAC 0      /* FLAC, 16 bits per sample */
#define ACE355_AUDIO_WAV_F32 1   /* RIFF/WAVE, IEEE float32 ("wav" and "wav32" of the reference) */
#define ACE355_AUDIO_WAV_PCM16 2 /* RIFF/WAVE, PCM_16 */

/* normalize_audio(audio, target_db), audio_utils.py:24-62, per item and in place: peak = max|x| over the item
 * ([channels, samples] jointly); items with peak < 1e-6 are left alone; x *= fp32(10^(dB/20)) * (1 / peak) with the
 * roundings torch applies.  wav dev f32 [n_items, per_item]; peaks_host (may be NULL) receives the peaks. */
int ace355_normalize_audio(float* wav_dev, int n_items, int64_t per_item, float target_db, float* peaks_host, void* stream);
/* dev f32 [n_items, channels, samples] -> dev interleaved [n_items, samples, channels]: int16 = lrintf(x * 32767)
 * saturated (libsndfile's float -> PCM_16 rule) when as_pcm16 != 0, else float32.  channels 1 or 2.  Asynchronous. */
int ace355_audio_interleave(const float* wav_dev, int n_items, int channels, int64_t samples, void* out_dev, int as_pcm16,
                            void* stream);
/* Host-side codecs (no GPU involved).  FLAC: RFC 9639 subset - fixed 4096-sample blocks, constant / verbatim /
 * fixed-predictor subframes, partitioned Rice coding, stereo decorrelation, CRC-8/16, MD5 in STREAMINFO; frames are
 * encoded by n_threads host threads (0 = up to 16).  pcm = interleaved int16 [frames, channels], channels 1 or 2. */
int64_t ace355_flac_bound(int64_t frames, int channels);
int ace355_flac_encode_pcm16(const int16_t* pcm, int64_t frames, int channels, int sample_rate, int n_threads, uint8_t* out,
                             int64_t cap, int64_t* out_len);
/* Decoder for AudioSaver.convert_audio (audio_utils.py:217-257) and round-trip checks: <= 16 bits per sample, 1-2
 * channels; constant / verbatim / fixed / LPC subframes, Rice and Rice2 residuals; checks every CRC, and the MD5
 * when verify_md5 != 0 and STREAMINFO carries one. */
int ace355_flac_info(const uint8_t* data, int64_t size, int64_t* frames, int32_t* channels, int32_t* sample_rate,
                     int32_t* bits_per_sample);
int ace355_flac_decode_pcm16(const uint8_t* data, int64_t size, int16_t* pcm_out, int64_t cap_samples, int verify_md5);
/* RIFF/WAVE: interleaved float32 (is_float != 0; format tag 3 + `fact` chunk) or int16 PCM. */
int64_t ace355_wav_bound(int64_t frames, int channels, int is_float);
int ace355_wav_encode(const void* interleaved, int64_t frames, int channels, int sample_rate, int is_float, uint8_t* out,
                      int64_t cap, int64_t* out_len);
/* AudioSaver.save_batch for a decoded batch still in HBM: interleave (+ quantise) on the GPU, one D2H copy of the
 * converted samples, all (item, block) encode jobs on one pool of n_threads host threads, files written in parallel.
 * wav dev f32 [n_items, channels, samples]; paths[n_items]; format = ACE355_AUDIO_*.  Synchronous. */
int ace355_save_audio_batch(const float* wav_dev, int n_items, int channels, int64_t samples, int sample_rate, int format,
                            const char* const* paths, int n_threads, void* stream);

/* ------------------------------------------------------------------------------------------
 * Unit kernels (test hooks; each has an oracle twin in oracle/ used by tests/)
 * ---------------------------------------------------------------------------------------- */
/* C = A[M,K] * W[N,K]^T, bf16 in, fp32 accumulate (MFMA).  out_dtype: ACE355_DTYPE_*; bias f32 [N] or NULL. */
int ace355_gemm_bf16(const void* A_dev, const void* W_dev, void* C_dev, int M, int N, int K, int out_dtype,
                     const float* bias_dev, void* stream);
/* Fused epilogues used by the DiT: mode 0: H[M,N] (f32) += gate * (A W^T), gate[n] = g1[n] + g2[(m / rows_per_seq)*g2_stride + n]
 * (NULL g1 -> gate 1);  mode 1: out[M,N/2] (bf16) = silu(gate) * up with W rows interleaved [32 gate | 32 up]. */
int ace355_gemm_bf16_fused(const void* A_dev, const void* W_dev, void* out_dev, int M, int N, int K, int mode,
                           const float* g1_dev, const float* g2_dev, int g2_stride, int rows_per_seq, void* stream);
/* Residual GEMM with the whole epilogue of the DiT's o_proj / down_proj launches (gemm.hip mode 2): H[M,N] (f32) += gate * (A W^T),
 * gate as in ace355_gemm_bf16_fused mode 0, plus cvec[n] on rows m >= cvec_row0 (the constant cross-attention term of the CFG
 * null branch, dit.hip forward_core).  cvec NULL: none. */
int ace355_gemm_bf16_residual(const void* A_dev, const void* W_dev, float* H_dev, int M, int N, int K, const float* g1_dev,
                              const float* g2_dev, int g2_stride, int rows_per_seq, const float* cvec_dev, int cvec_row0,
                              void* stream);
/* Projection GEMM with q / k head RMSNorm (+ RoPE when rope != 0) in its epilogue (gemm.hip mode 4; base.py:304, 338-343):
 * columns [0, q_cols) are q heads (norm weight wq [128]), [q_cols, qk_cols) k heads (wk), the rest pass through (v).
 * W bf16 [N,K] in the REFERENCE's row order; with rope the hook packs q / k rows into the library's head-pair order and the
 * q / k output columns of a head come back in that order too: dims (d, d+64) in columns (2d, 2d+1).  pos = row % rows_per_seq. */
int ace355_gemm_bf16_headnorm(const void* A_dev, const void* W_dev, void* out_bf16_dev, int M, int N, int K, int q_cols,
                              int qk_cols, const float* wq_dev, const float* wk_dev, float eps, int rope, int rows_per_seq,
                              float theta, void* stream);
/* OCP MXFP8 (gfx950 v_mfma_scale_f32_32x32x64_f8f6f4; BASELINE configs[4] "fp8 MFMA"; the reference's counterpart is torchao
 * fp8 on the DiT Linears, handler/init_service_loader.py:89-113).  mx_quantize: x bf16 [M,K] -> q fp8 e4m3 [M,K] + E8M0 block scales
 * (one per 32 consecutive K elements of a row) as uint32 [K/128][rows_pad]: byte b of word [kt][row] = block 4 kt + b;
 * rows_pad = ace355_mx_rows_pad(M).  gemm_mxfp8 (test hook): quantises both bf16 operands and runs the MX GEMM with the bf16 kernel's
 * epilogues: mode 0 out bf16 [M,N]; mode 2 H f32 [M,N] += gate * (A W^T) as ace355_gemm_bf16_fused mode 0; mode 3 SwiGLU -> bf16 [M,N/2].
 * K % 128 == 0, N % 256 == 0. */
int ace355_mx_quantize(const void* x_bf16_dev, int M, int K, void* q_out_dev, uint32_t* scales_out_dev, int rows_pad, void* stream);
int ace355_mx_rows_pad(int rows);
int ace355_gemm_mxfp8(const void* A_bf16_dev, const void* W_bf16_dev, void* out_dev, int M, int N, int K, int mode, const float* g1_dev,
                      const float* g2_dev, int g2_stride, int rows_per_seq, void* stream);
/* y = bf16( rmsnorm(x; w, eps) * (1 + sc) + sh ), sc[n] = sc1[n] + sc2[(m / rows_per_seq)*stride + n] (NULL -> no modulation). */
int ace355_rmsnorm_mod(const float* x_dev, const float* w_dev, void* y_bf16_dev, int M, int D, float eps,
                       const float* sc1, const float* sc2, const float* sh1, const float* sh2, int stride,
                       int rows_per_seq, void* stream);
/* In-place head RMSNorm (+ RoPE when rope != 0) on `heads` heads of 128 starting at column col0 of a bf16 [M, ld] matrix. */
int ace355_headnorm_rope(void* x_bf16_dev, int M, int ld, int col0, int heads, const float* w_dev, float eps,
                         int rope, int S, float theta, void* stream);
/* Flash attention: q bf16 [N,Sq,Hq*128] (ld = Hq*128), k bf16 [N,Skv,Hkv*128], v bf16 [N,Skv,Hkv*128];
 * window < 0 = full, else |i-j| <= window.  out bf16 [N,Sq,Hq*128]. */
int ace355_attention(const void* q_dev, const void* k_dev, const void* v_dev, void* out_dev, int N, int Sq, int Skv,
                     int Hq, int Hkv, int window, float scale, void* stream);
/* Same with the key-padding mask of the condition encoders (create_4d_mask, base.py:117-124, prefix form): keys
 * j >= kv_len_host[n] are masked; a query with no valid key attends uniformly to ALL keys, as the reference's
 * finfo.min additive mask does. */
int ace355_attention_masked(const void* q_dev, const void* k_dev, const void* v_dev, void* out_dev, int N, int Sq, int Skv,
                            int Hq, int Hkv, int window, float scale, const int32_t* kv_len_host, void* stream);
/* One guidance + Euler step (base.py:1946-1979): v dev f32 [2B,T,64] (cond | uncond), avg dev f32 [B,T,64]
 * momentum state (updated), xt dev f32 [B,T,64] (updated in place). first != 0: momentum buffer is empty. */
int ace355_apg_euler_step(const float* v_dev, float* avg_dev, float* xt_dev, int B, int T, float guidance, float dt,
                          int apply_cfg, int first, void* stream);
/* NLC conv (test hook for the VAE kernels): y[b,l,co] = bias[co] + sum_{tap,ci} f(x[b, l+(tap-center)*dil, ci]) * w[co,tap,ci]
 * (+ res[b,l,co]); f = snake(alpha,beta) when alpha != NULL.  x,y,res bf16 NLC; w bf16 [Cout,taps,Cin]. */
int ace355_conv1d_nlc(const void* x_dev, const void* w_dev, const float* bias_dev, const float* alpha_dev,
                      const float* beta_dev, const void* res_dev, void* y_dev, int B, int L, int Cin, int Cout,
                      int taps, int dilation, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ACE355_H */
