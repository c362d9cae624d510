/*
 * duodiff.h -- C ABI of libduodiff.so: the MI355X (gfx950) DuoDiff sampling engine.
 *
 * The reference (razvanmatisan/duodiff) is pure Python and has no FFI; its boundary
 * for the hot path is a pair of Python call signatures:
 *
 *     eps = model(x, time_tensor, y)            sampler.py:130-132, ddpm_core.py:150-152
 *     x   = postprocessing(eps, x, t)           sampler.py:133 -> sampler.py:47-56
 *
 * driven 1000 times by get_samples (sampler.py:82-155) with the backbone switch
 * at t == 1000 - t_switch (sampler.py:135-136).  This header is what a ctypes stub
 * on the reference side binds to replace exactly those calls.
 *
 * Conventions
 *   - plain C: pointers, sizes, ints.  No torch / HIP types in signatures; a stream is
 *     passed as void* (a hipStream_t; NULL = the null stream).
 *   - *_dev pointers are DEVICE pointers owned by the caller (e.g. torch-allocated).
 *     Images are fp32 NCHW contiguous, exactly like the reference's tensors.
 *   - every entry returns 0 on success, a negative dd_status on failure; the message is
 *     kept per context (dd_last_error).  No C++ exception and no abort() crosses the ABI.
 *   - a dd_ctx is bound to one device and is not re-entrant; all device work is enqueued
 *     asynchronously on the caller's stream, only dd_sync blocks.
 *   - there is NO CPU fallback: without a usable GPU dd_ctx_create fails.
 */
#ifndef DUODIFF_H
#define DUODIFF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_ABI_VERSION 5

typedef struct dd_ctx dd_ctx;
typedef struct dd_model dd_model;

typedef enum dd_status {
    DD_OK = 0,
    DD_ERR_INVALID = -1,      /* bad argument / shape mismatch   (Python: ValueError / RuntimeError) */
    DD_ERR_NOT_FOUND = -2,    /* unknown parameter name          (Python: KeyError)                  */
    DD_ERR_STATE = -3,        /* call order (e.g. forward before finalize)                          */
    DD_ERR_HIP = -4,          /* a HIP runtime call failed                                           */
    DD_ERR_NOMEM = -5,
    DD_ERR_UNSUPPORTED = -6   /* config outside what the kernels implement                          */
} dd_status;

/* U-ViT hyper-parameters: the YAML model_params block (reference configs/uvit_*.yaml,
 * constructor models/uvit.py:229-247). */
typedef struct dd_config {
    int32_t img_size;
    int32_t patch_size;
    int32_t in_chans;
    int32_t embed_dim;
    int32_t depth;
    int32_t num_heads;
    int32_t mlp_ratio;
    int32_t num_classes;          /* <= 0: unconditional */
    int32_t normalize_timesteps;  /* models/uvit.py:352-353 */
    int32_t max_batch;            /* activation workspace is sized for this many images */
    int32_t qkv_bias;             /* attn.qkv = nn.Linear(dim, 3 dim, bias=qkv_bias), models/uvit.py:150 (no shipped YAML sets it) */
    int32_t mlp_time_embed;       /* time_embed = Linear(D,4D) -> SiLU -> Linear(4D,D) on the sinusoid, models/uvit.py:264-272 */
} dd_config;

/* arithmetic mode of the GEMM / attention operands (accumulation, residual stream,
 * LayerNorm, softmax, head and DDPM update are fp32 in both) */
enum { DD_PREC_BF16 = 0, DD_PREC_FP32 = 1 };

/* sigma_t^2: beta-tilde as sampler.py:50, or beta as ddpm_core.py:57,72-75 (default there) */
enum { DD_VAR_BETA_TILDE = 0, DD_VAR_BETA = 1 };

/* where z ~ N(0, I) of the update comes from */
enum { DD_NOISE_NONE = 0, DD_NOISE_BUFFER = 1, DD_NOISE_PHILOX = 2 };

/* ---- context ---------------------------------------------------------------------- */
int dd_abi_version(void);
/* Hash of the sources this library was built from (duodiff_amd/build.py source_id): profile summaries record it, so a
 * counter value is only ever quoted for the build it was measured on.  Static string; needs no GPU. */
const char* dd_build_id(void);
int dd_ctx_create(int device, dd_ctx** out);
void dd_ctx_destroy(dd_ctx* ctx);
const char* dd_last_error(dd_ctx* ctx);            /* valid until the next call on ctx */
int dd_sync(dd_ctx* ctx, void* stream);

/* The five fp32[1000] schedule tables as the engine computes them (must be bit-equal to
 * sampler.py:40-44) plus the three per-step coefficients.  which: 0 betas, 1 alphas,
 * 2 alphas_bar, 3 alphas_bar_previous, 4 betas_tilde (sampler.py order),
 * 5 betas_tilde (ddpm_core.py:68-70 order), 6 c1=sqrt(1/alpha), 7 c2=(1-alpha)/sqrt(1-abar),
 * 8 sigma=sqrt(betas_tilde).  Works without a GPU (host arithmetic only). */
int dd_schedule_table(int which, float* out1000);

/* NoiseScheduler.__init__ (ddpm_core.py:56-70) for any (beta_init, beta_final, beta_steps): torch.linspace / cumprod
 * restated bit for bit; betas_tilde in the ddpm_core rounding order.  Each output holds `steps` floats or is NULL.
 * Host arithmetic only. */
int dd_schedule_build(float beta_init, float beta_final, int steps, float* betas, float* alphas, float* alphas_bar,
                      float* alphas_bar_prev, float* betas_tilde);

/* ---- model: replaces UViT(**model_params) + load_state_dict (sampler.py:271-293) ---- */
int dd_model_create(dd_ctx* ctx, const dd_config* cfg, dd_model** out);
/* name = the reference state_dict key (models/uvit.py:228-336), data = host fp32,
 * shape must match the reference's exactly.  Copies; may be called in any order. */
int dd_model_set_param(dd_model* m, const char* name, const float* host_data,
                       const int64_t* shape, int ndim);
/* all parameters present -> pack into kernel layouts on the device. */
int dd_model_finalize(dd_model* m, int precision);
int64_t dd_model_num_params(const dd_model* m);
void dd_model_destroy(dd_model* m);

/* ---- eps = model(x, t, y)  (models/uvit.py:351-383) --------------------------------- */
/* x_dev [B,C,S,S] fp32; t = the timestep every row of time_tensor holds (sampler.py:130);
 * t_dev: NULL, or the reference's time_tensor itself, [B] fp32 on the device, when rows differ
 * (then t is ignored); y_dev [B] int64 labels or NULL (must be non-NULL iff num_classes > 0,
 * quirk Q5); eps_dev [B,C,S,S] fp32 out. */
int dd_forward(dd_ctx* ctx, dd_model* m, const float* x_dev, float t, const float* t_dev,
               const int64_t* y_dev, float* eps_dev, int B, void* stream);

/* ---- early-exit baseline: EarlyExitUViT (models/early_exit.py:193-324) + eesampler.py ---- */
/* Which uncertainty probe the model carries (early_exit.py:194-204): the three MLPProbe tables (:31-37) or the
 * per-layer AttentionProbe (:40-80, the reference's default classifier_type). */
enum { DD_EE_MLP_PER_LAYER = 0, DD_EE_MLP_PER_TIMESTEP = 1, DD_EE_MLP_PER_LAYER_PER_TIMESTEP = 2, DD_EE_ATTENTION_PROBE = 3 };
/* Call right after dd_model_create.  dd_model_set_param then also takes the EarlyExitUViT state_dict names
 * (U-ViT names WITHOUT the "uvit." prefix, plus "matrix.<key>.classifier.0.{weight,bias}" -- for the attention probe
 * "matrix.<i>.{q, weight_kv.weight, weight_kv.bias, classification.0.weight, classification.0.bias,
 * classification.2.weight, classification.2.bias}" --,
 * "in_blocks_heads.<i>.*", "mid_block_head.*", "out_blocks_heads.<i>.*") and finalize requires all of them. */
int dd_model_enable_early_exit(dd_model* m, int classifier_type);
/* (eps, classifier_outputs, outputs) = EarlyExitUViT.forward(x, timesteps, y) (early_exit.py:270-320).
 * t = int(timesteps[0]) selects the probes; t_dev as in dd_forward.  classifier_dev [depth, B] fp32: the probe
 * of the input of every block; outputs_dev [depth, B, C, S, S] fp32: the OutputHead of the same inputs. */
int dd_forward_early_exit(dd_ctx* ctx, dd_model* m, const float* x_dev, float t, const float* t_dev,
                          const int64_t* y_dev, float* eps_dev, float* classifier_dev, float* outputs_dev,
                          int B, void* stream);
/* eesampler.py:61-71: per sample the first layer whose predicted error is <= threshold (the final output closes the
 * list with error 0; an all-False column selects layer 0 like torch.argmax) -> model_output_dev [B, chw],
 * indices_dev [B] int32 (or NULL), err_mean_dev [depth] = batch mean of classifier_dev rows (or NULL). */
int dd_early_exit_select(dd_ctx* ctx, const float* outputs_dev, const float* eps_dev, const float* classifier_dev,
                         float threshold, int depth, int B, int64_t chw, float* model_output_dev,
                         int32_t* indices_dev, float* err_mean_dev, void* stream);

/* ---- x = postprocessing(eps, x, t)  (sampler.py:47-56 == ddpm_core.py:190-193) ------ */
/* n = number of elements.  z_dev may be NULL (treated as 0); it is ignored when t == 0. */
int dd_ddpm_step(dd_ctx* ctx, const float* x_dev, const float* eps_dev, const float* z_dev,
                 int t, int variance, float* x_out_dev, int64_t n, void* stream);

/* The same update with the three scalars supplied by the caller: x' = c1 * (x - c2 * eps) + sigma * z (z_dev NULL: no
 * noise term), rounded op by op like the reference.  For schedules other than the 1000-step default
 * (NoiseScheduler(beta_steps=...), ddpm_core.py:167-193): the host derives c1 = sqrt(1/alpha_t),
 * c2 = (1-alpha_t)/sqrt(1-alphas_bar_t), sigma = sqrt(sigma_squared_t) from dd_schedule_build's tables. */
int dd_ddpm_step_coef(dd_ctx* ctx, const float* x_dev, const float* eps_dev, const float* z_dev, float c1, float c2,
                      float sigma, float* x_out_dev, int64_t n, void* stream);

/* ---- generic scalar-affine update: out = a*x + b*m + c*z (z_dev may be NULL) ---------- */
/* The form shared by predict_original / predict_previous post-processing (sampler.py:59-79) and a
 * DDIM step (sampler.py:112-120); the host computes a, b, c from the schedule tables. */
int dd_affine_step(dd_ctx* ctx, const float* x_dev, const float* m_dev, const float* z_dev, float a,
                   float b, float c, float* out_dev, int64_t n, void* stream);

/* ---- samples = rearrange((x + 1) / 2, "b c h w -> b h w c")  (sampler.py:145-146) -------- */
/* x_dev [B,C,S,S] fp32 -> images_dev [B,S,S,C] fp32, unclipped like the reference (quirk Q8).  Lets the whole
 * timed region of get_samples (sampler.py:327-345) stay inside the library: no torch op between the last step
 * and the D2H copy / the gather. */
int dd_to_images(dd_ctx* ctx, const float* x_dev, float* images_dev, int B, int C, int S, void* stream);

/* ---- one fused sampling step: x <- step(x, model(x,t,y), t) in place ---------------- */
/* noise_mode DD_NOISE_BUFFER: z_dev [B,C,S,S] supplies z (parity with the torch CPU stream);
 * DD_NOISE_PHILOX: z is generated on the device from (seed, t, element);
 * eps_out_dev (optional) also receives eps. */
int dd_sample_step(dd_ctx* ctx, dd_model* m, float* x_dev, int t, const int64_t* y_dev,
                   int noise_mode, const float* z_dev, uint64_t seed, int variance,
                   float* eps_out_dev, int B, void* stream);

/* ---- the whole loop: get_samples DDPM branch (sampler.py:128-139) ------------------- */
typedef struct dd_sample_args {
    dd_model* first;        /* model used from t = t_start                                    */
    dd_model* late;         /* or NULL; takes over AFTER the step at t == 1000 - t_switch     */
    int32_t t_switch;       /* <= 0: never switch (reference default: inf, quirk Q6)          */
    int32_t t_start;        /* normally 999                                                   */
    int32_t t_end;          /* normally 0 (inclusive)                                         */
    int32_t variance;       /* DD_VAR_*                                                       */
    int32_t noise_mode;     /* DD_NOISE_PHILOX or DD_NOISE_NONE (host noise: use dd_sample_step) */
    int32_t use_graph;      /* 1: capture one hipGraph per backbone and replay it             */
    uint64_t seed;
    const int64_t* y_dev;   /* [B] or NULL                                                    */
    float* x_dev;           /* in: x_T, out: x at t_end, [B,C,S,S] fp32                       */
    int32_t B;
    int32_t reserved;
} dd_sample_args;
int dd_sample(dd_ctx* ctx, const dd_sample_args* args, void* stream);

/* ---- the other loops of get_samples as device-resident loops: DDIM (sampler.py:103-126) and the predict_original /
 * predict_previous parametrisations (:59-79, :128-139).  Every one of them is  x' = a_k x + b_k model(x, t_k) + c_k z  per
 * step with host-computed scalars (duodiff_amd/sampler.py affine_coefficients, from the bit-exact schedule tables); the
 * table lives on the device, the step's last kernel applies row k and advances the device-resident step index, so one
 * captured hipGraph per backbone is replayed n_steps times (use_graph) -- or the same launches are issued eagerly, bit for bit
 * the same results.  z: device Philox (counter = step index) or none. */
typedef struct dd_affine_sample_args {
    dd_model* first;        /* model used from step 0                                                           */
    dd_model* late;         /* or NULL; runs from step index switch_after on                                    */
    int32_t n_steps;        /* number of updates                                                                */
    int32_t switch_after;   /* >= n_steps (or late == NULL): never                                              */
    const float* t;         /* host [n_steps]: the timestep the model sees at step k                            */
    const float* a;         /* host [n_steps]                                                                   */
    const float* b;         /* host [n_steps]                                                                   */
    const float* c;         /* host [n_steps]                                                                   */
    const int32_t* noise;   /* host [n_steps]: != 0: the c z term is added at step k (the reference skips it at its last step) */
    int32_t noise_mode;     /* DD_NOISE_PHILOX or DD_NOISE_NONE (c z never added)                               */
    int32_t use_graph;
    uint64_t seed;
    const int64_t* y_dev;   /* [B] or NULL                                                                      */
    float* x_dev;           /* in / out, [B,C,S,S] fp32                                                         */
    int32_t B;
    int32_t counter_base;   /* Philox counter of step 0 of THIS call: step k draws z with counter counter_base + k and key seed, so a loop cut
                             * into several calls (intermediate saves) with counter_base = steps already done draws the z of one uncut call   */
} dd_affine_sample_args;
int dd_sample_affine(dd_ctx* ctx, const dd_affine_sample_args* args, void* stream);

/* The early-exit baseline's loop (reference eesampler.py:40-89) as a device-resident loop: per step EarlyExitUViT.forward
 * (all heads and probes), the per-sample exit selection with the global threshold, the DDPM update (sigma^2 = beta-tilde)
 * with the selected output; row t of err_dev [1000, depth] (batch-mean predicted error per layer, :70) and of idx_dev
 * [1000, B] (exit layer per sample, :71) is written when the pointer is non-NULL.  One hipGraph for the step, replayed. */
typedef struct dd_ee_sample_args {
    dd_model* model;        /* created with dd_model_enable_early_exit                        */
    float threshold;
    int32_t t_start;        /* normally 999                                                   */
    int32_t t_end;          /* normally 0 (inclusive)                                         */
    int32_t noise_mode;     /* DD_NOISE_PHILOX or DD_NOISE_NONE                               */
    int32_t use_graph;
    int32_t B;
    uint64_t seed;
    const int64_t* y_dev;   /* [B] or NULL                                                    */
    float* x_dev;           /* in / out                                                       */
    float* err_dev;         /* [1000, depth] fp32 or NULL                                     */
    int32_t* idx_dev;       /* [1000, B] int32 or NULL                                        */
} dd_ee_sample_args;
int dd_sample_early_exit(dd_ctx* ctx, const dd_ee_sample_args* args, void* stream);

/* ---- KL-VAE decode: autoencoder.decode(x) at sampler.py:141-143 ---------------- */
/* Replaces FrozenAutoencoderKL.decode (models/utils/autoencoder.py:486-490, Decoder :320-449) for the fixed ddconfig of
 * get_autoencoder (:503-516).  Parameters are set by the reference state_dict keys ("decoder.*", "post_quant_conv.*";
 * "encoder.*" / "quant_conv.*" are accepted and ignored).  z_dev [B,4,h,h] fp32 latents -> out_dev [B,3,8h,8h] fp32. */
typedef struct dd_vae dd_vae;
int dd_vae_create(dd_ctx* ctx, int max_chunk, int max_latent, dd_vae** out);
int dd_vae_set_param(dd_vae* v, const char* name, const float* host_data, const int64_t* shape, int ndim);
int dd_vae_finalize(dd_vae* v, int precision);
int dd_vae_decode(dd_ctx* ctx, dd_vae* v, const float* z_dev, float* out_dev, int B, int latent_hw, void* stream);
void dd_vae_destroy(dd_vae* v);

/* ---- measurement support ------------------------------------------------------------ */
/* Time `iters` back-to-back launches of the fc1 GEMM (fused bias+GELU epilogue) of model m at batch B with hipEvents
 * on `stream`: the dominant kernel of the two-GEMM MLP path (fp32 mode, D > 512; bf16 models with D <= 512 run the fused
 * MLP kernel instead, timed in context by dd_profile_steps).  Returns average milliseconds per launch in *ms_out and
 * the launch's algorithmic FLOPs. */
int dd_bench_gemm(dd_ctx* ctx, dd_model* m, int B, int iters, void* stream,
                  float* ms_out, double* flops_out);
/* In-context timing of the dominant kernel: runs `steps` eager sampling steps (t = t_start, t_start-1, ...) in place
 * on x_dev with a hipEvent pair recorded on `stream` around EVERY launch of the block's dominant kernel (depth per step)
 * -- the fused block-tail kernel alone (attn.proj + norm2 + fc1 + GELU + fc2 + residual + next norm1 of the patch rows;
 * the small extra-token launches before and after it are OUTSIDE the pair) where the model uses it, else the fc1 GEMM --
 * and returns the average milliseconds per launch: what rocprofv3 --kernel-trace averages for that kernel. */
int dd_profile_steps(dd_ctx* ctx, dd_model* m, float* x_dev, const int64_t* y_dev, int t_start, int steps, int B,
                     void* stream, float* fc1_ms_out, int* launches_out);
/* Which launches the two profile entries bracket (default DD_PROF_DOMINANT; the choice stays with the context).  bench.py selects the
 * kernel with the largest total time in the workload's committed rocprofv3 kernel table (profiles/rNN/kernel_stats*.csv):
 *   DD_PROF_DOMINANT       the fused block tail where the model has one, else the mlp.fc1 GEMM (the round-1..4 behaviour)
 *   DD_PROF_BLOCK_TAIL     mlp_fused_kernel launches only
 *   DD_PROF_FC1            the mlp.fc1 GEMM (gemm256_kernel<bias + GELU>)
 *   DD_PROF_ROWLIN         every row-resident Linear launch (embed_dim 768: attn.proj, mlp.fc2, skip_linear -- one kernel, rowlin768_kernel)
 *   DD_PROF_QKV_ATTENTION  the attn.qkv + attention launch (qkv_attention_kernel), or the plain attention launch
 *   DD_PROF_SPLITK         every split-K GEMM launch (small-batch embed_dim 1024: attn.proj, mlp.fc2, skip_linear -- gemm256_kernel<partial>) */
#define DD_PROF_DOMINANT 0
#define DD_PROF_BLOCK_TAIL 1
#define DD_PROF_FC1 2
#define DD_PROF_ROWLIN 3
#define DD_PROF_QKV_ATTENTION 4
#define DD_PROF_SPLITK 5
int dd_profile_select(dd_ctx* ctx, int kind);
/* The same measurement for the way dd_sample runs an even batch >= 32: TWO half-batch chains (images [0, B/2) on `stream`, the rest on
 * the context's side stream), enqueued eagerly step by step, an event pair around every launch of the dominant kernel in both
 * chains -- each timed launch covers B/2 images and overlaps the other chain's kernels as in the timed loop. */
int dd_profile_steps_chained(dd_ctx* ctx, dd_model* m, float* x_dev, const int64_t* y_dev, int t_start, int steps, int B,
                             void* stream, float* ms_out, int* launches_out);

/* Host-only: the row partition the 256x256 GEMM uses for C[M,N] = A[M,K] W[N,K]^T on `num_cus` CUs:
 * q main tiles of 256 rows + e (<= 8) tail rows per tile; DD_ERR_UNSUPPORTED if the shape falls back
 * to the generic 128x128 kernel.  Needs no GPU. */
int dd_plan_rows(int M, int N, int K, int num_cus, int* q_out, int* e_out);

/* Number of CUs this context's persistent GEMM grids are sized for (default: the device's CU count, rounded down to
 * a multiple of 8; any value is rounded likewise).  For callers that run the engine on a CU-masked stream
 * (hipExtStreamCreateWithCUMask).  Graphs captured by dd_sample are keyed on the value: the next dd_sample re-captures.
 * A context (its step state, staging buffers and model workspaces) serves ONE stream at a time: calls on different
 * streams must be ordered by the caller. */
int dd_set_num_cus(dd_ctx* ctx, int num_cus);

/* Per-step timing of the last dd_sample call, measured with hipEvents on its stream:
 * [0] total ms, [1] ms in first-model steps, [2] ms in late-model steps. */
int dd_last_sample_timing(dd_ctx* ctx, float out3[3]);

#ifdef __cplusplus
}
#endif
#endif /* DUODIFF_H */
