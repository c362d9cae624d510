/*
 * duodiff_dev.h -- development / validation entry points of libduodiff.so.  NOT part of the drop-in boundary
 * (include/duodiff.h): nothing on the reference side binds these; tests and tools use them to drive one kernel alone.
 */
#ifndef DUODIFF_DEV_H
#define DUODIFF_DEV_H

#include "duodiff.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Development harness for the fused MLP kernel: x += fc2(gelu(fc1(bf16(h)) + b1)) + b2 on host arrays (h [M,D], nn.Linear
 * weights fp32, xres_host [M,D] in/out, out_host optional bf16 copy), plus `iters` timed launches.  extras == 0: the
 * rows are one image of M patch tokens; extras > 0: M / (1 + extras) images of `extras` extra tokens + 1 patch token.
 * ln_in [2, D] (gamma, beta): the kernel's fused-LayerNorm prologue computes h = LayerNorm(xres) itself (h_host is ignored);
 * ln_in NULL: h_host is the (already normalised) input;
 * ln_out [2, D] + ln_out_host [M, D] bf16 or NULL: LayerNorm of the updated rows from the epilogue.
 * ao_host [M, D] + wproj [D, D] + bproj [D] or NULL (needs ln_in, D % 128 == 0): the attention projection
 * x += ao . wproj^T + bproj runs in front of the MLP in the same launch (extra-token rows: the small kernel the model
 * launches for them).
 * skip_host [M, D] + wskip [D, 2D] + bskip [D] or NULL (needs the projection and ln_out): the NEXT block's skip_linear runs
 * behind the MLP in the same launch -- xres_host then receives x' = cat([y, skip]) . wskip^T + bskip instead of y, and
 * ln_out_host its LayerNorm (models/uvit.py:196-200, 206).
 * wqkv [3D, D] + qkv_out_host (bf16, HEAD-MAJOR: [M / L images][3 D / 64 units][Lp = L rounded up to 8][64], L = M for
 * extras == 0, else 1 + extras) or NULL (needs the projection and ln_out): the NEXT block's attn.qkv runs last in the same
 * launch on norm1 of the updated rows (models/uvit.py:152); ln_out_host is then only written for the extra-token rows. */
int dd_dev_mlp(dd_ctx* ctx, int M, int D, int hidden, int extras, const float* h_host, const float* w1, const float* b1,
               const float* w2, const float* b2, float* xres_host, unsigned short* out_host, const float* ln_in,
               const float* ln_out, unsigned short* ln_out_host, int iters, void* stream, float* ms_out,
               const float* ao_host, const float* wproj, const float* bproj, const float* skip_host, const float* wskip,
               const float* bskip, const float* wqkv, unsigned short* qkv_out_host);

/* Development harness for the attention launch that computes attn.qkv itself (attention.hip qkv_attention_kernel; bf16, 8 heads
 * of 64, L = 256 patches + `extras` = 1 or 2 leading extra tokens): out = softmax(q k^T / 8) v per (image, head) with
 * q, k, v = split(h . wqkv^T + bqkv), from host arrays h [B L, 512] (rounded to bf16), wqkv [1536, 512], bqkv [1536] or NULL;
 * out_host bf16 [B L, 512].  `iters` timed launches -> ms_out. */
int dd_dev_qkv_attention(dd_ctx* ctx, int B, int L, int H, int extras, const float* h_host, const float* wqkv, const float* bqkv,
                         unsigned short* out_host, int iters, void* stream, float* ms_out);

/* Development harness for the output head's first launch (rowops.hip head_dec_kernel; reference models/uvit.py:377-378):
 * dec = decoder_pred(LayerNorm(x)) in exact fp32 from host arrays x [M, D], norm gamma / beta [D], decoder_pred weight [pd, D] / bias [pd];
 * dec_host [M, pd] (rows the launch does not decode -- the first tok_e rows of every tok_l-row image when tok_l > 0 -- come back as NaN).
 * probe_w [D] + probe_b [1] + srow_host [M] or NULL (D = 256 / 512): the early-exit MLP probe's per-row value sigmoid(x . w + b) from the
 * same launch, for EVERY row (reference models/early_exit.py:31-37).  split != 0 (D = 256 / 512): the product as a split-bf16 product (hi + lo halves,
 * three bf16 MFMAs: 2^-16 of a product; what the bf16 engine's early-exit heads run) instead of the exact fp32 one.  `iters` timed launches -> ms_out. */
int dd_dev_head_dec(dd_ctx* ctx, int M, int D, int pd, int tok_l, int tok_e, const float* x_host, const float* norm_g, const float* norm_b,
                    const float* wdec, const float* bdec, float* dec_host, const float* probe_w, const float* probe_b, float* srow_host,
                    int split, int iters, void* stream, float* ms_out);

/* Kernel-variant switches for same-process A/B runs (tools/mlp_check.py, tools/all_configs.py).  They act on models
 * FINALIZED after the call (the first three) or on launches made after it; the product never sets them and the library
 * reads no environment variable. */
#define DD_DEV_NO_FUSED_MLP 1u      /* keep the fc1 / fc2 GEMM pair + LayerNorm launches instead of the fused block tail */
#define DD_DEV_NO_FUSED_PROJ 2u     /* keep attn.proj as its own GEMM */
#define DD_DEV_NO_FUSED_HEAD 4u     /* keep final LayerNorm + decoder_pred as two launches */
#define DD_DEV_NO_FUSED_SKIP 32u    /* keep skip_linear as its own GEMM + LayerNorm launch */
#define DD_DEV_NO_FUSED_QKV 64u     /* keep attn.qkv as its own GEMM launch */
#define DD_DEV_NO_FUSED_QA 128u     /* keep attn.qkv out of the attention launch (the qkv tensor goes through HBM) */
#define DD_DEV_GENERIC_EMBED 8u     /* generic VALU patch-embed kernel */
#define DD_DEV_MLP_EXTRAS_ONLY 16u  /* dd_dev_mlp: launch the hidden-split (extra-token) workgroups alone */
#define DD_DEV_NO_CHAINS 256u       /* dd_sample: one chain over the whole batch (default: two half-batch chains on two streams for even B >= 32) */
#define DD_DEV_NO_ROWLIN 1024u       /* embed_dim 768: keep mlp.fc2 as a GEMM + LayerNorm launch pair (default: the row-resident launch of rowlin.hip) */
#define DD_DEV_NO_ROWLIN_PROJ 2048u  /* embed_dim 768: keep attn.proj as a GEMM + LayerNorm launch pair */
#define DD_DEV_NO_EMBED_LN 4096u     /* keep the first block's norm1 as its own launch (default: written by the patch-embed launch where it fits) */
#define DD_DEV_NO_SPLITK 8192u       /* small-batch GEMM-path models: keep skip_linear / attn.proj / mlp.fc2 as whole-K GEMMs + LayerNorm launches */
#define DD_DEV_NO_ROWLIN_SKIP 16384u /* embed_dim 768: keep the out-blocks' skip_linear as a GEMM + LayerNorm launch pair */
#define DD_DEV_NO_SPLIT_HEADS 32768u /* early-exit heads of the bf16 engine: keep the exact-fp32 decoder product (default: the split-bf16 product at embed_dim 256 / 512) */
#define DD_DEV_FORCE_CHAINS 512u    /* dd_sample: two half-batch chains for ANY even batch (tests at small batches) */
int dd_dev_set_flags(dd_ctx* ctx, unsigned flags);

/* Number of hipGraph captures dd_sample has made on this context so far (tests: a second call with other tensors of the
 * same shape must not capture again). */
long long dd_dev_graph_captures(dd_ctx* ctx);

/* Fill every activation buffer of model m -- the main workspace and the second chain's (allocated here if dd_sample has not yet) -- with
 * 0xFF bytes (NaN as bf16 and as fp32) ON `stream`.  Tests: a run that follows must equal a run on a fresh model bit for bit, i.e. no kernel
 * may depend on what a buffer held before the launches of the call wrote it (zero-initialised padding, stale slabs). */
int dd_dev_poison_workspaces(dd_ctx* ctx, dd_model* m, void* stream);

/* Number of chains the last dd_sample call on this context ran (1, or 2 half-batch chains on two streams). */
int dd_dev_last_sample_chains(dd_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* DUODIFF_DEV_H */
