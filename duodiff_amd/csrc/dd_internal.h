// Internal declarations shared by the HIP translation units of libduodiff.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dd {

typedef unsigned short bf16_t;  // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    return __builtin_bit_cast(float, ((unsigned)b) << 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kPerChunk = 4;  // elements per 16-byte chunk
    __device__ static __forceinline__ float from_f32(float v) { return v; }
    __device__ static __forceinline__ float to_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int kPerChunk = 8;
    __device__ static __forceinline__ bf16_t from_f32(float v) { return f2bf(v); }
    __device__ static __forceinline__ float to_f32(bf16_t v) { return bf2f(v); }
};

// Device-resident per-step state, so that a captured hipGraph replays without parameter
// patching: kernels read t from here, the last node of a step decrements it.
struct StepState {
    int t;            // current timestep (999 .. 0): read by the FIRST kernel of a step (token assembly) only
    float t_model;    // what the model sees: t or t/1000 is derived in the embed kernel
    unsigned long long seed;
    int t_final;      // copy of t made by the embed kernel and read by the LAST kernel of the step, which may then
                      // decrement t / t_model for the next replay without racing its own blocks (no separate launch)
    int pad;
};

// per-timestep update coefficients, x' = c1 (x - c2 eps) + sigma z
struct StepCoef {
    float c1, c2, sigma_tilde, sigma_beta;
};

// one step of a table-driven loop (dd_sample_affine): x' = a x + b eps + c z, the model sees t_model
struct AffineRow {
    float t_model, a, b, c;
    int noise, ctr, pad1, pad2;   // ctr: the Philox counter of this step's z (the step's index in the WHOLE loop, not in this call's table)
};

enum GemmEpilogue {
    EPI_STORE = 0,        // out = T(acc)                                  (qkv)
    EPI_BIAS_GELU = 1,    // out = T(gelu_erf(acc + bias))                 (fc1)
    EPI_BIAS_RESID = 2,   // x += acc + bias ; optional out = T(x)         (proj, fc2)
    EPI_BIAS_SET = 3,     // x  = acc + bias                               (skip_linear)
    EPI_BIAS_STORE = 4,   // out = T(acc + bias)                           (VAE attention q/k projections)
    EPI_PARTIAL = 5       // partial[split] = acc over the split's k range (launch_gemm_splitk; launch_reduce_ln finishes the Linear)
};

// Head-major output map of the qkv Linear (what attention.hip reads): element (row m = b * L + l, column c) of the [M, 3D]
// result lives at ((b * 3H + (c >> 6)) * Lp + l) * 64 + (c & 63) -- the 64 columns of one (q | k | v, head) unit of one image
// are Lp contiguous 128-byte (bf16) rows, so a GEMM wave stores whole 4 KB blocks and the attention workgroup stages its K / V
// with linear LDS-DMA.  Lp = L rounded up to 8 (one LDS-DMA piece = 8 rows); rows l >= L are never written (the workspace
// is zeroed once).  L == 0: plain [M, ldo] rows.
struct HeadMajor {
    int L, Lp, H;
    unsigned magic;     // 2^32 / L + 1:  b = umulhi(m, magic)  (exact for m * L < 2^32)
};
inline HeadMajor make_head_major(int L, int H) {
    return HeadMajor{L, (L + 7) / 8 * 8, H, (unsigned)((1ull << 32) / (unsigned)L + 1ull)};
}
__device__ __forceinline__ long long hm_offset(const HeadMajor& hm, int row, int col) {
    const int b = (int)__umulhi((unsigned)row, hm.magic), l = row - b * hm.L;
    return (((long long)b * (3 * hm.H) + (col >> 6)) * hm.Lp + l) * 64 + (col & 63);
}

// C[M,N] = [A | A2][M,K] . W[N,K]^T ; A holds k < K1, A2 holds the rest (concat-free skip GEMM).
template <typename T>
struct GemmArgs {
    const T* A;
    const T* A2;
    const T* W;        // [N, K] row-major (nn.Linear layout)
    const float* bias; // [N] or null
    float* xres;       // fp32 [Mp, N] residual stream (EPI_BIAS_RESID / EPI_BIAS_SET)
    T* out;            // [Mp, ldo] or null
    int M, N, K, K1;
    int lda, lda2, ldo;
    HeadMajor hm;      // hm.L != 0: `out` is written head-major (the qkv Linear); N % 64 == 0
    float* partial = nullptr;   // launch_gemm_splitk: fp32 slabs [splits][M][N]
    int splits = 0;
    int tile128 = -1;           // bf16 launch_gemm: 1 = the 128 x 128 kernel, 0 = the 256 x 256 kernel where the shape fits it, -1 = decided from THIS call's shape
                                // (gemm_prefers_128).  The model's launches pass the decision made for its max_batch, so that a row's kernel never
                                // depends on the batch of a call (the two half-batch chains of dd_sample must compute what the whole batch computes)
};

// num_cus: CU count the persistent bf16 grid is sized for (per context; a multiple of 8)
template <typename T> hipError_t launch_gemm(const GemmArgs<T>& a, int epilogue, hipStream_t s, int num_cus);
bool plan_rows_256(int M, int N, int K, int num_cus, int* q, int* e);
bool gemm_prefers_128(int M, int N, int K, int K1, int num_cus);   // the 256 x 256 tiles of this shape would leave half of the CUs without one
bool gemm_splitk_supported(int M, int N, int K, int K1, int splits);
hipError_t launch_gemm_splitk(const GemmArgs<bf16_t>& a, hipStream_t s, int num_cus);
int device_num_cus();

// ---- fused MLP (mlp_fused.hip): x += fc2(gelu(fc1(h) + b1)) + b2 in one launch
struct MlpFusedArgs {
    const bf16_t* X;       // [Mp, ldx] LayerNorm output h (bf16); unused when ln_in_g is set
    int ldx;
    const float* ln_in_g;  // set: the kernel computes LayerNorm(x) * ln_in_g + ln_in_b itself from the residual rows
    const float* ln_in_b;  //      (the W1 image must then be packed with kperm = true)
    const bf16_t* ao;      // nproj > 0: attention output rows [Mp, D]; the kernel adds ao . Wproj^T + bproj to x first
    const float* bproj;    //            (main tiles only: the extra-token rows must already hold x1)
    int nproj;             // D / 32 blocks of Wproj in front of the MLP blocks of wimg, or 0
    const bf16_t* skip;    // nskip > 0: the NEXT block's skip_linear runs behind the MLP (main tiles): x' = [y | skip] . Wskip^T + bskip
    const float* bskip;    //            replaces y in xres, ln_out = norm1(x'); skip = the long-skip operand's rows [Mp, D]
    int nskip;             // 2 * D / 32 blocks of Wskip behind the MLP blocks of wimg (mlp_fused_pack_skip), or 0
    bf16_t* qkv_out;       // nqkv > 0: the NEXT block's attn.qkv (no bias) runs last (main tiles): qkv = norm1(updated rows) . Wqkv^T, written
    bf16_t* ln_out_frag;   // or null: the main tiles write norm1 HERE instead of ln_out, in MFMA fragment order ([32-row group of patch rows]
                           //          [D / 16 k-steps][64 lanes] x 16 bytes) for launch_qkv_attention; needs tok_n % 32 == 0
    bf16_t* qkv_dump;      //           head-major (hm) -- norm1 is then NOT written for the main rows; qkv_dump: 16 KB scratch (qkv | x | bf16 copy of rows past the end)
    HeadMajor hm;          //           (stores of rows past the end of a ragged tile)
    int nqkv;              // 3 * D / 32 blocks of Wqkv closing the image (mlp_fused_pack_rows), or 0
    const float* ln_out_g; // with ln_out: LayerNorm of the UPDATED rows, written as bf16 [Mp, D] (next block's norm1)
    const float* ln_out_b;
    bf16_t* ln_out;
    const char* wimg;      // weight image in MFMA fragment order (mlp_fused_pack)
    const float* b1p;      // fc1 bias, permuted to accumulator-register order per chunk
    const float* b2;       // [D]
    float* xres;           // fp32 [Mp, D] residual stream, updated in place
    bf16_t* out;           // optional bf16 copy of the updated rows, row stride ldo
    int ldo;
    float* partial;        // partial slabs of the hidden-split leftover tiles
    int nchunks;           // hidden / 32
    // row plan (mlp_fused_plan): token row of image b, token l = b * tok_l + l; tokens [0, tok_e) are the extras, then tok_n patches
    int tok_n, tok_e, tok_l;
    int n_main, n_extra;   // B * tok_n patch rows (main tiles), B * tok_e extra rows (hidden-split tiles)
    int tiles_main, tiles_left, groups, cpg;
    int prows;             // rows per hidden-split tile (32, 64 or 128: whole waves)
    float* y_tap = nullptr; // SKIP launches of early-exit models: the block output y (fp32 [Mp, D]) is stored here before skip_linear replaces it in xres -- the
                           // patch rows by the main tiles, the extra-token rows by the reduce launch (the next block's output head and probe read y)
    int reduce_set = 0;    // launch_mlp_reduce: x = b2 + slabs instead of x += (the extra-token rows of a row-resident skip_linear; of a block tail with the
                           // projection in front: the first hidden group's slab carries x + proj(ao) + b)
};
bool mlp_fused_supported(int D, int hidden);
size_t mlp_fused_image_bytes(int D, int hidden, bool with_proj, bool with_skip, bool with_qkv);
// nn.Linear weight [nrows, D] -> nrows / 32 blocks (one 32-row tile each, fragment f = k-step f, k index in accumulator order)
void mlp_fused_pack_rows(int D, int nrows, const float* w, unsigned short (*to_bf16)(float), unsigned short* img);
// qkv of the extra-token rows of a QKV launch (from a.ln_out, which the reduce / skip_rows launch has written for them)
hipError_t launch_qkv_rows(const MlpFusedArgs& a, int D, hipStream_t s);
void mlp_fused_pack_proj(int D, const float* wp, unsigned short (*to_bf16)(float), unsigned short* img);
void mlp_fused_pack_skip(int D, const float* ws, unsigned short (*to_bf16)(float), unsigned short* img);
hipError_t launch_skip_rows_ln(const MlpFusedArgs& a, int D, hipStream_t s, bool with_ln = true);   // with_ln = false: x' only, column-split (D = 512)
size_t mlp_fused_partial_bytes(int max_batch, int extras, int D, int hidden);
void mlp_fused_plan(int B, int n_patches, int extras, int seq_len, int hidden, MlpFusedArgs& a);
void mlp_fused_pack(int D, int hidden, const float* w1, const float* b1, const float* w2, bool kperm,
                    unsigned short (*to_bf16)(float), unsigned short* img, float* b1p);
hipError_t launch_mlp_fused(const MlpFusedArgs& a, int D, hipStream_t s);
hipError_t launch_mlp_reduce(const MlpFusedArgs& a, int D, hipStream_t s);
hipError_t init_mlp_fused_kernels();

// ---- row-resident Linear + bias + residual + LayerNorm for embed_dim 768 (rowlin.hip): x += A . W^T + b ; LayerNorm(x) g + beta as bf16
struct RowLinArgs {
    const bf16_t* A;       // [rows, lda] bf16 rows of the Linear's input (k contiguous)
    const bf16_t* A2;      // k_split > 0: the input is cat[A | A2] along k (skip_linear): k >= k_split comes from A2 [rows, lda]
    int k_split;           // a multiple of 64, or 0
    int set_x;             // != 0: x = A . W^T + b (no residual: skip_linear), else x += ...
    int lda, K;            // K % 64 == 0
    const char* wimg;      // rowlin_pack image of W [768, K]
    const float* bias;     // [768]
    float* xres;           // fp32 [rows, 768] residual stream, updated in place
    const float *ln_g, *ln_b;   // LayerNorm of the updated rows (h_out / h_frag)
    bf16_t* h_out;         // row-major [rows, 768] bf16, or null
    bf16_t* h_frag;        // or: the main (patch) rows in MFMA fragment order (MlpFusedArgs::ln_out_frag), or null
    bf16_t* x_copy;        // optional bf16 copy of the updated rows, row-major
    float* partial;        // slabs of the K-split extra-token tiles [tiles_extra * groups][128][768] fp32
    int M;                 // plain mode (tok_n == 0): rows [0, M) in tiles of 128
    // row plan (rowlin_plan; tok_n > 0): token row of image b, token l = b * tok_l + l; tokens [0, tok_e) are the extras, then tok_n patches
    int tok_n, tok_e, tok_l, n_main, n_extra, tiles_main, tiles_extra, groups, cpg;
};
bool rowlin_supported(int D, int K);
void rowlin_pack(int K, const float* w, unsigned short (*to_bf16)(float), unsigned short* img);
void rowlin_plan(int B, int n_patches, int extras, int seq_len, int K, RowLinArgs& a);
size_t rowlin_partial_bytes(int max_batch, int extras, int K);
hipError_t launch_rowlin(const RowLinArgs& a, hipStream_t s);
hipError_t init_rowlin_kernels();

struct EmbedArgs {
    const float* x_img;      // [B,C,S,S]
    const float* wt;         // [pd, D]  patch-embed weight, transposed
    const float* bias;       // [D]
    const float* pos;        // [L, D]
    const float* label_emb;  // [num_classes, D] or null
    const long long* y;      // [B] or null
    const float* t_vec;      // [B] per-row timesteps or null (then st->t_model)
    StepState* st;           // t_model is read, t_final is written (block 0)
    float* x_tok;            // [Mp, D]
    int B, C, S, P, D, L, extras, num_classes, normalize, Mp;
    int generic;             // != 0: the generic VALU kernel even where the MFMA kernel fits (development A/B runs)
    const float* ln_g = nullptr;   // with ln_frag: the first block's norm1 of the patch rows is written too (embed_ln_supported)
    const float* ln_b = nullptr;
    bf16_t* ln_frag = nullptr;     // MFMA fragment order (launch_layernorm_frag's `frag`)
};
bool embed_ln_supported(const EmbedArgs& a);

// finishes a split-K Linear (launch_gemm_splitk) and runs the LayerNorm behind it: rowops.hip reduce_ln_kernel
struct ReduceLnArgs {
    float* x;               // fp32 [rows, D] residual stream (read when resid != 0, written always)
    const float* partial;   // fp32 slabs [splits][slab elements], row-major [rows, D] each
    long long slab;         // elements per slab (the GEMM's M * N)
    int splits, resid;
    const float* bias;      // [D]
    bf16_t* copy;           // optional bf16 copy of the updated rows, row stride ldo
    int ldo;
    const float *ln_g, *ln_b;   // optional LayerNorm of the updated rows ...
    bf16_t* h;              // ... row-major [rows, D] (with frag: only the extra-token rows go here)
    bf16_t* frag;           // ... or: the patch rows in MFMA fragment order (launch_layernorm_frag's `frag`)
    int tok_l, tok_e;
    int rows;
};
hipError_t launch_reduce_ln(const ReduceLnArgs& a, int D, hipStream_t s);
hipError_t launch_embed(const EmbedArgs& a, hipStream_t s);

// time_embed MLP (models/uvit.py:264-272): time token = W2 . SiLU(W1 . sinusoid(t) + b1) + b2 (+ pos_embed), fp32, one
// workgroup per image; overwrites the time-token row the embed kernel wrote.  w1t [D, 4D], w2t [4D, D]: transposed on the host.
struct TimeMlpArgs {
    const float *w1t, *b1, *w2t, *b2, *pos, *t_vec;
    const StepState* st;
    float* x_tok;
    int B, D, L, extras, normalize;
};
hipError_t launch_time_mlp(const TimeMlpArgs& a, hipStream_t s);

template <typename T>
hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, T* out,
                            int rows, int D, hipStream_t s);
hipError_t launch_layernorm_frag(const float* x, const float* gamma, const float* beta, bf16_t* out, bf16_t* frag, int rows, int D,
                                 int tok_l, int tok_e, hipStream_t s);   // patch rows in MFMA fragment order (MlpFusedArgs::ln_out_frag)

// qkv: head-major (HeadMajor, make_head_major(L, H)); out: [B * L, D] rows
template <typename T>
hipError_t launch_attention(const T* qkv, T* out, int B, int L, int H, int D, hipStream_t s);
// attn.qkv + attention in one launch (attention.hip qkv_attention_kernel): h = norm1 of the patch rows in fragment order
// (MlpFusedArgs::ln_out_frag); the extra-token rows: hx = norm1 row-major [B L, D], or hx = nullptr and xres / ln_g / ln_b = the fp32
// residual stream and this block's norm1 parameters (the kernel normalises the rows itself); wimg from qkv_attention_pack;
// bf16, D = 512, L = 256 + extras only
bool qkv_attention_supported(int D, int H, int L, int extras);
void qkv_attention_pack(int D, int H, const float* w, unsigned short (*to_bf16)(float), unsigned short* img);
hipError_t launch_qkv_attention(const bf16_t* h, const bf16_t* wimg, const float* bias, const bf16_t* hx, const float* xres,
                                const float* ln_g, const float* ln_b, bf16_t* out, int B, int L, int H, int D, int extras, hipStream_t s);

struct FinalArgs {
    const float* dec;      // [B*L, pd] decoder_pred output for every token (extras included)
    const float* wconv;    // [C, C, 3, 3]
    const float* bconv;    // [C]
    const float* x_in;     // [B,C,S,S] or null (forward only)
    const float* z;        // [B,C,S,S] or null
    float* eps_out;        // or null
    float* x_out;          // or null (may alias x_in)
    StepState* st;         // reads t_final / seed; advance != 0: one thread sets t = t_final - 1 for the next step
    const StepCoef* coef;  // [1000]
    int B, C, S, P, L, extras, noise_mode, variance;
    int advance;
    const AffineRow* atab; // or null.  Set: st->t is a STEP INDEX k into this table; the update is a x + b eps + c z with row k,
                           // and advance hands row k + 1's timestep to the next step (the table holds one row more than steps)
    int b0 = 0;            // index of this launch's first image within the whole batch (a half-batch chain of dd_sample): only the
                           // Philox pixel ids depend on it, so that a chain draws the z the undivided batch would
    // (the fields below are set by name, never positionally: the aggregate initialisers of the step launches end at b0)
    int layer_B = 0;       // > 0: the B "images" are layer_B images of B / layer_B early-exit layers, one after the other (dec, eps_out contiguous that way);
    long long w_stride = 0, b_stride = 0;   //   layer i convolves with wconv + i * w_stride / bconv + i * b_stride (floats): ONE launch for every layer's head
};
hipError_t launch_final(const FinalArgs& a, hipStream_t s);

// final LayerNorm + decoder_pred in one launch (rowops.hip); wg = decoder weight * norm gamma [pd, D], c = decoder bias + W . norm beta
struct HeadDecArgs {
    const float* x;    // [Mp, D] fp32 residual stream
    const float* wg;   // [pd, D]
    const float* c;    // [2 pd]: c, then the row sums of wg (the kernel multiplies the un-normalised rows: rowops.hip)
    float* dec;        // [Mp, pd]
    int M, pd;
    int tok_l, tok_e;  // > 0: rows are images of tok_l tokens whose first tok_e (the extra tokens) are NOT decoded; 0: every row
    // early-exit MLP probe folded into the launch (srow != null; D = 256 / 512): srow[row] = sigmoid(x[row,:] . w + b) for EVERY row (the extra
    // tokens' too), w = pw_base + pi D, b = pb_base[pi], pi = st->t_final * t_mul + add (launch_ee_probe's row selection)
    float* srow = nullptr;
    const float* pw_base = nullptr;
    const float* pb_base = nullptr;
    const StepState* st = nullptr;
    int t_mul = 0, add = 0;
    int split = 0;     // 1: wg is the packed split-bf16 image (pack_head_split) and c's row sums are of hi + lo; D = 256 / 512 (rowops.hip: SPLIT)
};
bool head_dec_probe_supported(int D);
// Wg [pd, D] -> the SPLIT kernel's image: [jj < D / 32][ct < nt][hi, lo][64 lanes] x 8 bf16 (2 D nt 16 bf16 = the fp32 matrix's bytes at pd = 16 nt);
// wsum [pd]: the row sums of hi + lo (what the kernel multiplies)
void pack_head_split(int D, int pd, const float* wg, unsigned short (*f2bf)(float), unsigned short* img, float* wsum);
bool head_dec_supported(int D, int pd);
hipError_t launch_head_dec(const HeadDecArgs& a, int D, int num_cus, hipStream_t s);

hipError_t launch_ddpm_step(const float* x, const float* eps, const float* z, float* out,
                            StepCoef c, int use_noise, long long n, hipStream_t s);
template <typename T> hipError_t launch_fill_random(T* p, long long n, unsigned seed, float scale, hipStream_t s);
// ---- KL-VAE decoder bandwidth kernels (vae_kernels.hip)
hipError_t launch_vae_input(const float* z, const float* w, const float* b, float inv_scale, float* out, int B, int HW, hipStream_t s);
hipError_t launch_vae_output(const float* in, float* out, int B, int C, int HW, int ldc, hipStream_t s);
template <typename T> hipError_t launch_im2col3x3(const T* src, T* dst, int B, int H, int W, int C, int up, int Kpad, hipStream_t s);
template <typename T> hipError_t launch_im2col3x3_c4(const float* src, T* dst, int B, int H, int W, int Kpad, hipStream_t s);
template <typename T> hipError_t launch_groupnorm(const float* x, float* part, const float* gamma, const float* beta, T* out, int B, int HW, int C, int swish, hipStream_t s);
int groupnorm_partials(int B, int HW);
template <typename T> hipError_t launch_softmax_rows(const float* sc, T* p, long long rows, int n, float scale, hipStream_t s);
template <typename T> hipError_t launch_cast(const float* x, T* out, long long n, hipStream_t s);

hipError_t launch_affine_step(const float* x, const float* m, const float* z, float* out, float a, float b, float c,
                              long long n, hipStream_t s);
// AttentionProbe operands of one layer (capi.hip finalize folds them): u [D], Wv^T [D, D], bv [D], W0^T [D, D], b0 [D], w2 [D], b2 [1]
struct AttnProbeW { const float *u, *wvt, *bv, *w0t, *b0, *w2, *b2; };
hipError_t launch_ee_attn_probe(const float* x, const AttnProbeW& w, float* out, int B, int L, int D, hipStream_t s);
// probe row = (t_mul ? st->t_final * t_mul : 0) + add of the [n_probe, D] / [n_probe] tables; srow: [B, L] fp32 scratch (the rows' sigmoids)
hipError_t launch_ee_probe(const float* x, const float* w_base, const float* bias_base, float* out, float* srow, int B, int L, int D,
                           const StepState* st, int t_mul, int add, hipStream_t s);
hipError_t launch_ee_probe_reduce(const float* srow, float* out, int rows, int L, hipStream_t s);   // out == nullptr above: the rows only; this finishes any number of (layer, image) rows
// st != null: idx / err_mean are [1000, B] / [1000, depth] tables and row st->t_final is written
// a half-batch chain (dd_sample_early_exit): B = the chain's images, idx rows are idx_stride = the whole batch wide and the chain writes from
// column idx_col0 on; sums: err_mean receives the plain per-layer SUM over the chain's images (launch_ee_mean_combine joins the chains)
hipError_t launch_ee_select(const float* outs, const float* eps, const float* cls, float thr, int depth, int B, long long chw,
                            float* mo, int* idx, float* err_mean, const StepState* st, hipStream_t s, int idx_stride = 0, int idx_col0 = 0, bool sums = false);
// launch_ee_select + launch_ddpm_step_state in one launch (the device-resident loop; idx_col0 = the chain's first image within the whole batch)
hipError_t launch_ee_select_step(float* x, const float* outs, const float* eps, const float* cls, float thr, int depth, int* idx, float* err_mean,
                                 int idx_stride, int idx_col0, bool sums, StepState* st, const StepCoef* coef, int B, int C, int S,
                                 int noise_mode, int advance, hipStream_t s);
hipError_t launch_ee_mean_combine(const float* s0, const float* s1, float* err, int depth, int t_lo, int t_hi, int B, hipStream_t s);
// b0: index of the launch's first image within the whole batch (only the Philox pixel ids depend on it)
hipError_t launch_ddpm_step_state(float* x, const float* eps, StepState* st, const StepCoef* coef, int B, int C, int S,
                                  int noise_mode, int advance, hipStream_t s, int b0 = 0);
hipError_t launch_set_state(StepState* st, int t, unsigned long long seed, hipStream_t s);
hipError_t launch_set_state_table(StepState* st, const AffineRow* atab, unsigned long long seed, hipStream_t s);   // step index 0
hipError_t launch_set_state_float(StepState* st, float t, hipStream_t s);
// (x + 1) / 2, NCHW -> NHWC: the output convention of reference sampler.py:145-146
hipError_t launch_to_images(const float* x, float* out, int B, int C, int S, hipStream_t s);

}  // namespace dd
