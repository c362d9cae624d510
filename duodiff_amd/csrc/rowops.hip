// Bandwidth-bound kernels around the GEMM/attention core (gfx950): token assembly,
// LayerNorm, the output head, the 3x3 conv fused with the DDPM update, device noise.
// All arithmetic here is fp32 in both precision modes.
#include "dd_internal.h"

#include <cstring>
#include <utility>
#include <vector>

#include <cstdlib>

namespace dd {
namespace {

// ------------------------------------------------------------------------------------------
// Token assembly: reference models/uvit.py:352-365 + PatchEmbed :221-225 + timestep_embedding :95-115
//   row order per image: [label_emb[y]] , time token , N patch tokens ; + pos_embed
// One workgroup builds TOK token rows of one image; thread d-strided over the embedding dim so
// every store and every read of W^T / pos_embed is coalesced.  Rows >= B*L (padding) are zeroed.
// ------------------------------------------------------------------------------------------
constexpr int kEmbedTok = 8;

__global__ void __launch_bounds__(256) embed_kernel(const EmbedArgs a) {
    __shared__ float patch[kEmbedTok][64];  // pd <= 64
    const int rows_per_img_blocks = (a.L + kEmbedTok - 1) / kEmbedTok;
    const int b = blockIdx.x / rows_per_img_blocks;
    const int r0 = (blockIdx.x % rows_per_img_blocks) * kEmbedTok;
    const int tid = threadIdx.x;
    const int pd = a.C * a.P * a.P;
    const int g = a.S / a.P;
    if (blockIdx.x == 0 && tid == 0) a.st->t_final = a.st->t;   // handed to the step's last kernel (StepState)

    if (b >= a.B) {  // padding rows of the workspace
        const long long row0 = (long long)a.B * a.L + (long long)(blockIdx.x - a.B * rows_per_img_blocks) * kEmbedTok;
        for (int j = 0; j < kEmbedTok; ++j) {
            const long long row = row0 + j;
            if (row < a.Mp)
                for (int d = tid; d < a.D; d += 256) a.x_tok[row * a.D + d] = 0.f;
        }
        return;
    }

    // gather the pixels of the patch rows handled here: k = c*P*P + p1*P + p2 (conv weight order)
    for (int idx = tid; idx < kEmbedTok * pd; idx += 256) {
        const int j = idx / pd, k = idx % pd;
        const int row = r0 + j;
        float v = 0.f;
        if (row >= a.extras && row < a.L) {
            const int n = row - a.extras;
            const int gy = n / g, gx = n % g;
            const int c = k / (a.P * a.P), p1 = (k / a.P) % a.P, p2 = k % a.P;
            v = a.x_img[(((long long)b * a.C + c) * a.S + gy * a.P + p1) * a.S + gx * a.P + p2];
        }
        patch[j][k] = v;
    }
    __syncthreads();

    const float t_raw = a.t_vec ? a.t_vec[b] : a.st->t_model;
    const float tt = a.normalize ? t_raw / 1000.0f : t_raw;
    const int halfd = a.D / 2;

    for (int d = tid; d < a.D; d += 256) {
        float acc[kEmbedTok];
#pragma unroll
        for (int j = 0; j < kEmbedTok; ++j) acc[j] = 0.f;
        for (int k = 0; k < pd; ++k) {
            const float w = a.wt[(long long)k * a.D + d];
#pragma unroll
            for (int j = 0; j < kEmbedTok; ++j) acc[j] = fmaf(w, patch[j][k], acc[j]);
        }
        const float bias = a.bias[d];
#pragma unroll
        for (int j = 0; j < kEmbedTok; ++j) {
            const int row = r0 + j;
            if (row >= a.L) continue;
            float v;
            if (row >= a.extras) {
                v = acc[j] + bias;
            } else if (row == a.extras - 1) {
                // sinusoidal time token: [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(1e4) * i / half)
                const int i = d < halfd ? d : d - halfd;
                const float f = expf((-9.210340371976184f * (float)i) / (float)halfd);
                const float arg = tt * f;
                v = (d < halfd) ? cosf(arg) : (d < 2 * halfd ? sinf(arg) : 0.f);
            } else {
                long long yy = a.y[b];
                yy = yy < 0 ? 0 : (yy >= a.num_classes ? a.num_classes - 1 : yy);
                v = a.label_emb[yy * a.D + d];
            }
            a.x_tok[((long long)b * a.L + row) * a.D + d] = v + a.pos[(long long)row * a.D + d];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Patch embedding on the matrix pipe (exact fp32: v_mfma_f32_16x16x4_f32 == an fmaf chain in k order), for image grids
// of 16 patches per row (every shipped config; other grids take the generic kernel above).  A VALU kernel is bound by its
// PD * D FMAs per token (0.8 G per launch = 26 us of pure vector issue -- round 1's embed_fast_kernel, 68 us); the f32 MFMA
// does the same arithmetic at twice that rate and leaves the vector pipe to the address math.
//   workgroup = 8 waves = 8 patch rows of an image (2 workgroups per image: one per CU at B = 128, a single round);
//   wave      = one patch row = 16 tokens as the MFMA N dimension: B operand = the 16 patches (lane (n, kq): pixel k =
//               4 kk + kq of token n -- per kk one coalesced read of a whole image row segment), A operand = the
//               transposed conv weight from LDS, 16 embedding columns per tile, two tiles (one 128-byte line of every
//               token) per loop iteration.
// The extra tokens (time sinusoid, label embedding) of an image are written by the workgroup that owns its first patch rows,
// one column per thread.
// ------------------------------------------------------------------------------------------
// LND > 0 (= embed_dim, compile time): the wave also writes the first block's norm1 of its 16 tokens -- in the MFMA fragment order the attention
// launch loads (layernorm_kernel's `frag`) -- from the embedded rows it still holds in registers: two-pass statistics over the four lanes
// of a token, no LayerNorm launch, no second read of x.
template <int P, int C, int LND = 0>
__global__ void __launch_bounds__(512) embed_mfma_kernel(const EmbedArgs a) {
    constexpr int PD = P * P * C, KK = PD / 4, KK4 = (KK + 3) / 4;
    static_assert(PD % 4 == 0, "k-steps of 4");
    extern __shared__ __attribute__((aligned(16))) char emb_lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(emb_lds);            // [D / 32][2][KK4][64 lanes] x 4 k-steps
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int D = a.D, S = a.S, NP = D / 32;
    // gridDim.y > 1 (few images: ImageNet-256 latents at B = 32 are 64 workgroups): the column pairs are dealt over blockIdx.y, each
    // workgroup parks and multiplies only its share of the weight image (same arithmetic per element)
    const int NPY = NP / (int)gridDim.y, p0 = (int)blockIdx.y * NPY;
    const int pass = blockIdx.x * 8 + wave, b = pass >> 4, gy = pass & 15;
    const bool valid = b < a.B;
    if (blockIdx.x == 0 && tid == 0) a.st->t_final = a.st->t;   // handed to the step's last kernel (StepState)

    // B operand: pixel k = 4 kk + kq of token n (k = (c, p1, p2) as in the reference's Conv2d weight [D, C, P, P])
    float pv[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const int k = 4 * kk + kq, c = k / (P * P), p1 = (k / P) % P, p2 = k % P;
        pv[kk] = valid ? a.x_img[(((long long)b * C + c) * S + gy * P + p1) * S + n * P + p2] : 0.f;
    }
    // A operand image -> LDS, all loads of a thread before its first write
    {
        const int items = NPY * 2 * KK4 * 64;
        constexpr int MAXI = 16;                               // items per thread: D = 1024, PD = 16 -> 8; D = 768, PD = 48 -> 18 (two rounds)
        for (int base = 0; base < items; base += 512 * MAXI) {
            f32x4 v[MAXI];
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int item = base + tid + 512 * i;
                v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (item < items) {
                    const int l = item & 63, kk4 = (item >> 6) % KK4, ph = (item >> 6) / KK4;   // ph = 2 p + half
                    const int col = 16 * (ph + 2 * p0) + (l & 15);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int kk = 4 * kk4 + j;
                        if (kk < KK) v[i][j] = a.wt[(long long)(4 * kk + (l >> 4)) * D + col];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int item = base + tid + 512 * i;
                if (item < items) wl[item] = v[i];
            }
        }
    }
    __syncthreads();
    if (valid) {
        const long long l0 = a.extras + gy * 16 + n;
        float* xrow = a.x_tok + ((long long)b * a.L + l0) * D + 4 * kq;
        const float* prow = a.pos + l0 * D + 4 * kq;
        const float* brow = a.bias + 4 * kq;
        const f32x4* wlp = wl + lane;
        // pos_embed / bias quads of pair p + 1 are requested before pair p computes (a load behind every iteration's
        // stores would expose one L2 round trip per pair).  The loop is unrolled by two with two register sets: rotating one set
        // (cur = next; next = load) compiles into copies of the freshly loaded registers at the END of the iteration that issued
        // the loads -- a vmcnt(0) per pair, i.e. exactly the exposed round trip the prefetch is there to hide (2 us x 16 pairs).
        struct Quads { f32x4 pe, po, be, bo; };
        auto fetch = [&](int pp) -> Quads {
            return Quads{*reinterpret_cast<const f32x4*>(prow + 32 * pp), *reinterpret_cast<const f32x4*>(prow + 32 * pp + 16),
                         *reinterpret_cast<const f32x4*>(brow + 32 * pp), *reinterpret_cast<const f32x4*>(brow + 32 * pp + 16)};
        };
        auto pair = [&](int pp, const Quads& q) {
            f32x4 acc_e = {0.f, 0.f, 0.f, 0.f}, acc_o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk4 = 0; kk4 < KK4; ++kk4) {
                const f32x4 we = wlp[((2 * (pp - p0)) * KK4 + kk4) * 64], wo = wlp[((2 * (pp - p0) + 1) * KK4 + kk4) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (4 * kk4 + j < KK) {
                        acc_e = __builtin_amdgcn_mfma_f32_16x16x4f32(we[j], pv[4 * kk4 + j], acc_e, 0, 0, 0);
                        acc_o = __builtin_amdgcn_mfma_f32_16x16x4f32(wo[j], pv[4 * kk4 + j], acc_o, 0, 0, 0);
                    }
                }
            }
            // lane (n, kq) holds columns 32 p + 4 kq + {0..3} (even tile) and 32 p + 16 + 4 kq + {0..3} (odd tile) of token n:
            // the four lanes of a token write 64 contiguous bytes per store, the two stores one 128-byte line
            const f32x4 ve = (acc_e + q.be) + q.pe, vo = (acc_o + q.bo) + q.po;
            *reinterpret_cast<f32x4*>(xrow + 32 * pp) = ve;
            *reinterpret_cast<f32x4*>(xrow + 32 * pp + 16) = vo;
            return std::pair<f32x4, f32x4>{ve, vo};
        };
        Quads qa = fetch(p0), qb = qa;
        if constexpr (LND > 0) {
            constexpr int NPC = LND / 32;
            f32x4 keep[2 * NPC];                  // the token's columns 16 t + 4 kq .. + 3, t = 0 .. D / 16 - 1
#pragma unroll
            for (int p = 0; p < NPC; p += 2) {
                qb = fetch(p + 1);
                const auto r0 = pair(p, qa);
                keep[2 * p] = r0.first; keep[2 * p + 1] = r0.second;
                qa = fetch(p + 2 < NPC ? p + 2 : p + 1);
                const auto r1 = pair(p + 1, qb);
                keep[2 * p + 2] = r1.first; keep[2 * p + 3] = r1.second;
            }
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2 * NPC; ++t) s4 += keep[t];
            float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float mean = sum / (float)LND;
            f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 2 * NPC; ++t) { const f32x4 d = keep[t] - mean; q4 += d * d; }
            float q2 = (q4[0] + q4[1]) + (q4[2] + q4[3]);
            q2 += __shfl_xor(q2, 16);
            q2 += __shfl_xor(q2, 32);
            const float rstd = 1.0f / sqrtf(q2 / (float)LND + 1e-5f);
            // fragment order: [32-row group of patch rows][k-step][64 lanes] x 16 bytes; columns 16 t + 4 kq .. + 3 of the token = k-step t,
            // lane (row in group) + 32 (kq >> 1), elements 4 (kq & 1) .. + 3
            const long long grp = (long long)b * 8 + (gy >> 1);
            bf16_t* frow = a.ln_frag + ((grp * (LND / 16)) * 64 + ((gy & 1) * 16 + n) + 32 * (kq >> 1)) * 8 + 4 * (kq & 1);
            const float* gr = a.ln_g + 4 * kq;
            const float* br = a.ln_b + 4 * kq;
#pragma unroll
            for (int t = 0; t < 2 * NPC; ++t) {
                const f32x4 gq = *reinterpret_cast<const f32x4*>(gr + 16 * t), bq = *reinterpret_cast<const f32x4*>(br + 16 * t);
                bf16_t o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = f2bf((keep[t][e] - mean) * rstd * gq[e] + bq[e]);
                *reinterpret_cast<uint2*>(frow + (long long)t * 512) = *reinterpret_cast<const uint2*>(o4);
            }
        } else {
            for (int p = p0; p < p0 + NPY; p += 2) {        // (NPY is even: launch_embed)
                qb = fetch(p + 1);
                pair(p, qa);
                qa = fetch(p + 2 < p0 + NPY ? p + 2 : p + 1);
                pair(p + 1, qb);
            }
        }
    }
    if (blockIdx.y == 0 && ((blockIdx.x * 8) & 15) == 0 && (int)(blockIdx.x * 8) / 16 < a.B) {   // the image's extra tokens: [label,] time (reference models/uvit.py:356-365)
        const int bb = (blockIdx.x * 8) >> 4;
        const float t_raw = a.t_vec ? a.t_vec[bb] : a.st->t_model;
        const float tt = a.normalize ? t_raw / 1000.0f : t_raw;
        const int halfd = D / 2;
        for (int row = 0; row < a.extras; ++row)
            for (int d = tid; d < D; d += 512) {
                float v;
                if (row == a.extras - 1) {
                    const int i = d < halfd ? d : d - halfd;
                    const float arg = tt * expf((-9.210340371976184f * (float)i) / (float)halfd);
                    v = d < halfd ? cosf(arg) : (d < 2 * halfd ? sinf(arg) : 0.f);
                } else {
                    long long yy = a.y[bb];
                    yy = yy < 0 ? 0 : (yy >= a.num_classes ? a.num_classes - 1 : yy);
                    v = a.label_emb[yy * D + d];
                }
                a.x_tok[((long long)bb * a.L + row) * D + d] = v + a.pos[(long long)row * D + d];
            }
    }
    if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) {   // padding rows of the workspace
        const long long first = (long long)a.B * a.L * D, count = ((long long)a.Mp - (long long)a.B * a.L) * D;
        for (long long i = tid; i < count; i += 512) a.x_tok[first + i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// time_embed MLP of mlp_time_embed=True models (reference models/uvit.py:264-272, applied at :358):
//   time_token = Linear(4D, D)( SiLU( Linear(D, 4D)( timestep_embedding(t, D) ) ) )
// One workgroup per image, fp32 throughout; the transposed weights make every read coalesced.  B * 16 D^2 FLOPs per
// step -- nothing next to a block -- so this stays a plain VALU kernel.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) time_mlp_kernel(const TimeMlpArgs a) {
    extern __shared__ float tm_lds[];            // [D] sinusoid, then [4D] hidden
    float* e = tm_lds;
    float* hmid = tm_lds + a.D;
    const int b = blockIdx.x, tid = threadIdx.x, D = a.D, H4 = 4 * a.D, halfd = a.D / 2;
    const float t_raw = a.t_vec ? a.t_vec[b] : a.st->t_model;
    const float tt = a.normalize ? t_raw / 1000.0f : t_raw;
    for (int d = tid; d < D; d += 256) {         // same sinusoid as the embed kernels: [cos(t f_i) | sin(t f_i)]
        const int i = d < halfd ? d : d - halfd;
        const float arg = tt * expf((-9.210340371976184f * (float)i) / (float)halfd);
        e[d] = d < halfd ? cosf(arg) : (d < 2 * halfd ? sinf(arg) : 0.f);
    }
    __syncthreads();
    for (int j = tid; j < H4; j += 256) {
        float acc = a.b1[j];
        for (int k = 0; k < D; ++k) acc = fmaf(a.w1t[(long long)k * H4 + j], e[k], acc);
        hmid[j] = acc / (1.0f + expf(-acc));     // SiLU
    }
    __syncthreads();
    float* row = a.x_tok + ((long long)b * a.L + a.extras - 1) * D;
    for (int d = tid; d < D; d += 256) {
        float acc = a.b2[d];
        for (int k = 0; k < H4; ++k) acc = fmaf(a.w2t[(long long)k * D + d], hmid[k], acc);
        row[d] = acc + a.pos[(long long)(a.extras - 1) * D + d];
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm (eps 1e-5, biased variance, affine): one wave per token row, row held in registers,
// two-pass statistics.  fp32 in, T out (the GEMM operand type).
// ------------------------------------------------------------------------------------------
constexpr int kLnMaxPerLane = 16;  // D <= 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// frag != nullptr (bf16, D % 256 == 0, tok_n % 32 == 0): the patch rows (token l >= tok_e of every tok_l-row image) are written
// THERE in MFMA B-fragment order -- [32-row group of patch rows][D / 16 k-steps][64 lanes] x 16 bytes, what the attention launch
// that computes attn.qkv itself loads (MlpFusedArgs::ln_out_frag) -- instead of row-major; the extra-token rows still go to `out`.
template <typename T>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out,
                                                        int rows, int D, T* __restrict__ frag = nullptr, int tok_l = 0, int tok_e = 0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long long)row * D;
    float v[kLnMaxPerLane];
    const int per = D / 64;  // D is a multiple of 64
    float s = 0.f;
    if ((D & 255) == 0) {
        // vector path: lane owns float4 chunks lane + 64*j
#pragma unroll
        for (int j = 0; j < kLnMaxPerLane / 4; ++j) {
            if (j * 256 < D) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(xr + j * 256 + lane * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[j * 4 + e] = q[e]; s += q[e]; }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kLnMaxPerLane; ++j)
            if (j < per) { v[j] = xr[j * 64 + lane]; s += v[j]; }
    }
    const float mean = wave_sum(s) / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < kLnMaxPerLane; ++j)
        if (j < per) { const float dlt = v[j] - mean; q2 += dlt * dlt; }
    const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)D + 1e-5f);
    T* orow = out + (long long)row * D;
    bool to_frag = false;
    if constexpr (sizeof(T) == 2) {
        if (frag) {
            const int b = row / tok_l, l = row - b * tok_l;
            if (l >= tok_e) {
                const int n = l - tok_e, grp = b * ((tok_l - tok_e) / 32) + n / 32;
                orow = frag + ((long long)grp * (D / 16) * 64 + (n & 31)) * 8;
                to_frag = true;
            }
        }
    }
    if ((D & 255) == 0) {
#pragma unroll
        for (int j = 0; j < kLnMaxPerLane / 4; ++j) {
            if (j * 256 < D) {
                const int c0 = j * 256 + lane * 4;
                const f32x4 gq = *reinterpret_cast<const f32x4*>(gamma + c0);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(beta + c0);
                T o4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o4[e] = Elem<T>::from_f32((v[j * 4 + e] - mean) * rstd * gq[e] + bq[e]);
                if constexpr (sizeof(T) == 2) {
                    // fragment order: columns c0 .. c0+3 = k-step c0 / 16, lane half (c0 / 8) & 1, elements c0 & 7 ..
                    const long long off = to_frag ? ((long long)(c0 >> 4) * 64 + 32 * ((c0 >> 3) & 1)) * 8 + (c0 & 7) : c0;
                    *reinterpret_cast<uint2*>(orow + off) = *reinterpret_cast<const uint2*>(o4);
                } else {
                    *reinterpret_cast<f32x4*>(orow + c0) = *reinterpret_cast<const f32x4*>(o4);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kLnMaxPerLane; ++j)
            if (j < per) {
                const int c = j * 64 + lane;
                orow[c] = Elem<T>::from_f32((v[j] - mean) * rstd * gamma[c] + beta[c]);
            }
    }
}

// ------------------------------------------------------------------------------------------
// Second half of a split-K Linear (launch_gemm_splitk) and the LayerNorm behind it in one row pass: one wave per token row,
//   acc = ((p_0 + p_1) + ...) + bias   (ascending split order);   x = resid ? x + acc : acc;   [bf16 copy of x];   [LayerNorm(x) g + b]
// The LayerNorm launch that followed the Linear read x anyway; here it reads the slabs too and the GEMM's own epilogue is gone.
// Two-pass statistics and output orders (row-major / fragment order for the patch rows) as layernorm_kernel.
// ------------------------------------------------------------------------------------------
template <int NQ>      // D = 256 NQ
__global__ void __launch_bounds__(256) reduce_ln_kernel(const ReduceLnArgs a) {
    constexpr int D = 256 * NQ;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    float* xr = a.x + (long long)row * D + lane * 4;
    f32x4 v[NQ], pv[NQ], bq[NQ], xv[NQ];
    const float* pr = a.partial + (long long)row * D + lane * 4;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        v[j] = *reinterpret_cast<const f32x4*>(pr + 256 * j);
        bq[j] = *reinterpret_cast<const f32x4*>(a.bias + 256 * j + lane * 4);
        if (a.resid) xv[j] = *reinterpret_cast<const f32x4*>(xr + 256 * j);
    }
    for (int sp = 1; sp < a.splits; ++sp) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) pv[j] = *reinterpret_cast<const f32x4*>(pr + (long long)sp * a.slab + 256 * j);
#pragma unroll
        for (int j = 0; j < NQ; ++j) v[j] += pv[j];
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        v[j] += bq[j];
        if (a.resid) v[j] = xv[j] + v[j];
        *reinterpret_cast<f32x4*>(xr + 256 * j) = v[j];
    }
    if (a.copy) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            bf16_t o4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = f2bf(v[j][e]);
            *reinterpret_cast<uint2*>(a.copy + (long long)row * a.ldo + 256 * j + lane * 4) = *reinterpret_cast<const uint2*>(o4);
        }
    }
    if (!a.ln_g) return;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j) sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    const float mean = wave_sum(sum) / (float)D;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[j][e] - mean; q2 += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)D + 1e-5f);
    bf16_t* orow = a.h + (long long)row * D;
    bool to_frag = false;
    if (a.frag) {
        const int b = row / a.tok_l, l = row - b * a.tok_l;
        if (l >= a.tok_e) {
            const int n = l - a.tok_e, grp = b * ((a.tok_l - a.tok_e) / 32) + n / 32;
            orow = a.frag + ((long long)grp * (D / 16) * 64 + (n & 31)) * 8;
            to_frag = true;
        }
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const int c0 = 256 * j + lane * 4;
        const f32x4 gq = *reinterpret_cast<const f32x4*>(a.ln_g + c0), lb = *reinterpret_cast<const f32x4*>(a.ln_b + c0);
        bf16_t o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = f2bf((v[j][e] - mean) * rstd * gq[e] + lb[e]);
        const long long off = to_frag ? ((long long)(c0 >> 4) * 64 + 32 * ((c0 >> 3) & 1)) * 8 + (c0 & 7) : c0;
        *reinterpret_cast<uint2*>(orow + off) = *reinterpret_cast<const uint2*>(o4);
    }
}

// ------------------------------------------------------------------------------------------
// Device noise: Philox4x32-10 counter RNG + Box-Muller.  Counter = (element/4, t, 0, 0),
// key = seed.  Statistically N(0,1); NOT the torch CPU mt19937 stream (that is DD_NOISE_BUFFER).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3,
                                             unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// four N(0,1) values for one pixel (one per channel, C <= 4): counter = (pixel, t), key = seed
__device__ __forceinline__ f32x4 philox_normal4(unsigned long long seed, unsigned long long pixel, int t) {
    unsigned c0 = (unsigned)pixel, c1 = (unsigned)(pixel >> 32), c2 = (unsigned)t, c3 = 0x5eedu;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // Box-Muller on (c0,c1) and (c2,c3); hardware log/sin/cos (|error| ~1e-6) are ample for noise
    const float ua = ((float)c0 + 1.0f) * 2.3283064365386963e-10f, ub = (float)c1 * 2.3283064365386963e-10f;
    const float uc = ((float)c2 + 1.0f) * 2.3283064365386963e-10f, ud = (float)c3 * 2.3283064365386963e-10f;
    const float ra = sqrtf(-2.0f * __logf(ua)), rc = sqrtf(-2.0f * __logf(uc));
    const float aa = 6.283185307179586f * ub, ac = 6.283185307179586f * ud;
    return f32x4{ra * __cosf(aa), ra * __sinf(aa), rc * __cosf(ac), rc * __sinf(ac)};
}

// ------------------------------------------------------------------------------------------
// Output head (after final LayerNorm + decoder_pred GEMM) + DDPM update: unpatchify ("B (h w) (p1 p2 C) -> B C (h p1) (w p2)",
// reference models/uvit.py:125-132), 3x3 conv pad 1 (:382), then
//   x <- sqrt(1/a_t) (x - (1-a_t)/sqrt(1-abar_t) eps) + sigma_t z      (sampler.py:47-56)
// One thread per pixel, all output channels; eps never goes to HBM unless asked for.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) final_kernel(const FinalArgs a) {
#pragma clang fp contract(off)  // the update must round like the reference: mul, sub, mul, add -- no FMA
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const int S = a.S, P = a.P, C = a.C;
    const long long npix = (long long)a.B * S * S;
    if (pix >= npix) return;
    const int b = (int)(pix / (S * S)), y = (int)((pix / S) % S), x = (int)(pix % S);
    const int g = S / P, pd = P * P * C;

    float acc[4];  // C <= 4
#pragma unroll
    for (int co = 0; co < 4; ++co) acc[co] = co < C ? a.bconv[co] : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= S) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int xx = x + dx - 1;
            if (xx < 0 || xx >= S) continue;
            const long long tok = (long long)b * a.L + a.extras + (yy / P) * g + (xx / P);
            const float* u = a.dec + tok * pd + ((yy % P) * P + (xx % P)) * C;
            for (int ci = 0; ci < C; ++ci) {
                const float uv = u[ci];
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    if (co < C) acc[co] = fmaf(a.wconv[((co * C + ci) * 3 + dy) * 3 + dx], uv, acc[co]);
            }
        }
    }

    const int t = a.st->t_final;
    if (a.atab) {   // table-driven loop (dd_sample_affine): t is the step index
        const AffineRow row = a.atab[t];
        if (a.advance && pix == 0) { a.st->t = t + 1; a.st->t_model = a.atab[t + 1].t_model; }
        const bool nz = row.noise != 0 && a.noise_mode == 2;
        f32x4 zn = {0.f, 0.f, 0.f, 0.f};
        if (a.x_out && nz) zn = philox_normal4(a.st->seed, (unsigned long long)(pix + (long long)a.b0 * S * S), row.ctr);
        for (int co = 0; co < C; ++co) {
            const long long e = (((long long)b * C + co) * S + y) * S + x;
            const float eps = acc[co];
            if (a.eps_out) a.eps_out[e] = eps;
            if (a.x_out) {
                float v = row.a * a.x_in[e] + row.b * eps;      // affine_step_kernel's order and roundings
                if (nz) v = v + row.c * zn[co];
                a.x_out[e] = v;
            }
        }
        return;
    }
    if (a.advance && pix == 0) { a.st->t = t - 1; a.st->t_model = (float)(t - 1); }   // no block of this kernel reads t / t_model
    const StepCoef cf = a.coef[t < 0 ? 0 : (t > 999 ? 999 : t)];
    const float sigma = a.variance == 1 ? cf.sigma_beta : cf.sigma_tilde;
    f32x4 zn = {0.f, 0.f, 0.f, 0.f};
    if (a.x_out && t > 0 && a.noise_mode == 2) zn = philox_normal4(a.st->seed, (unsigned long long)(pix + (long long)a.b0 * S * S), t);
    for (int co = 0; co < C; ++co) {
        const long long e = (((long long)b * C + co) * S + y) * S + x;
        const float eps = acc[co];
        if (a.eps_out) a.eps_out[e] = eps;
        if (a.x_out) {
            // same operation order and roundings as the reference (no FMA contraction)
            float v = cf.c1 * (a.x_in[e] - cf.c2 * eps);
            if (t > 0) {
                if (a.noise_mode == 1) v = v + sigma * a.z[e];
                else if (a.noise_mode == 2) v = v + sigma * zn[co];
            }
            a.x_out[e] = v;
        }
    }
}

// Tiled variant of final_kernel: a workgroup owns a 16x16 pixel tile of one image and first parks the
// 18x18xC halo of the unpatchified decoder output in LDS, so every decoder value is fetched once
// instead of up to nine times.  Same arithmetic, same rounding order (a tap outside the image multiplies a zero of the halo instead of
// being skipped: fmaf(w, 0, acc) == acc).
// CT = the channel count at compile time (3, 4; 0: a.C at run time): with it the 9 C^2 conv weights are unconditional scalar loads, one
// output channel's 9 C at a time, PT = the patch size likewise (the halo gather divides by it) -- the run-time form tested co < C / ci < C around every one of 144 candidate loads (221 scalar branches,
// 151 s_load_dword, the weights' SGPRs spilled to VGPR lanes: ~3 900 instructions for a kernel every sampling step waits for).
template <int CT, int PT>
__global__ void __launch_bounds__(256) final_tiled_kernel(const FinalArgs a) {
#pragma clang fp contract(off)
    // (row pitch 48 = 16 mod 32 words: the two 16-pixel rows a 32-lane group reads fall into disjoint bank halves; pitch 19 gave every tap read
    // a 2-way conflict on three banks -- 1.7 conflict cycles per LDS-active cycle in the round 2-4 profiles, for a kernel that is latency, not LDS)
    __shared__ float u[4][18][48];
    const int S = a.S, P = PT ? PT : a.P, C = CT ? CT : a.C, g = S / P, pd = P * P * C;
    const int tiles = (S + 15) / 16;
    const int b = blockIdx.x / (tiles * tiles), ty = (blockIdx.x / tiles) % tiles, tx = blockIdx.x % tiles;
    const int tid = threadIdx.x;
    const int ly = tid >> 4, lx = tid & 15, y = ty * 16 + ly, x = tx * 16 + lx;
    const bool inside = y < S && x < S;
    // Everything that does not depend on the decoder output is requested / computed first, so that its latency (step
    // state -> coefficient row, the x_t pixels, the Philox normals) overlaps the halo gather instead of following it:
    // the kernel is a chain of dependent memory round trips, not bandwidth.
    const int t = a.st->t_final;
    const bool table = a.atab != nullptr;      // dd_sample_affine: t is a step index, the update a x + b eps + c z of row t
    const AffineRow row = table ? a.atab[t] : AffineRow{0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0};
    const float t_next = (table && a.advance) ? a.atab[t + 1].t_model : 0.f;
    const StepCoef cf = a.coef[table ? 0 : (t < 0 ? 0 : (t > 999 ? 999 : t))];
    const bool draw = table ? (row.noise != 0) : (t > 0);
    float xin[4] = {0.f, 0.f, 0.f, 0.f}, zin[4] = {0.f, 0.f, 0.f, 0.f};
    if (inside && a.x_out) {
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            if (co < C) {
                const long long e = (((long long)b * C + co) * S + y) * S + x;
                xin[co] = a.x_in[e];
                if (draw && a.noise_mode == 1) zin[co] = a.z[e];
            }
        }
    }
    const int layer = a.layer_B > 0 ? b / a.layer_B : 0;      // (early-exit heads batched into one launch: this image's layer)
    const float* wconv = a.wconv + layer * a.w_stride;
    const float* bconv = a.bconv + layer * a.b_stride;
    for (int idx = tid; idx < 18 * 18; idx += 256) {
        const int hy = idx / 18, hx = idx % 18;
        const int yy = ty * 16 + hy - 1, xx = tx * 16 + hx - 1;
        const bool in = yy >= 0 && yy < S && xx >= 0 && xx < S;
        const float* src = a.dec + ((long long)b * a.L + a.extras + (in ? (yy / P) * g + (xx / P) : 0)) * pd +
                           (in ? ((yy % P) * P + (xx % P)) * C : 0);
        for (int ci = 0; ci < C; ++ci) u[ci][hy][hx] = in ? src[ci] : 0.f;
    }
    f32x4 zn = {0.f, 0.f, 0.f, 0.f};
    if (inside && a.x_out && draw && a.noise_mode == 2)
        zn = philox_normal4(a.st->seed, ((unsigned long long)(b + a.b0) * S + y) * S + x, table ? row.ctr : t);   // same pixel id as the untiled kernel (b0: this launch's first image within the whole batch)
    __syncthreads();
    if (a.advance && blockIdx.x == 0 && tid == 0) {   // no block of this kernel reads t / t_model
        const int tn = table ? t + 1 : t - 1;
        a.st->t = tn;
        a.st->t_model = table ? t_next : (float)tn;
    }
    if (!inside) return;
    constexpr int CM = CT ? CT : 4;
    float uu[CM][9];
#pragma unroll
    for (int ci = 0; ci < CM; ++ci)
#pragma unroll
        for (int k = 0; k < 9; ++k) uu[ci][k] = ci < C ? u[ci][ly + k / 3][lx + k % 3] : 0.f;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int co = 0; co < CM; ++co) {
        if (co < C) {
            // uniform address, constant address space: scalar loads (s_load_dwordx8 ...), one output channel's 9 C weights live at a time
            const __attribute__((address_space(4))) float* wp = (const __attribute__((address_space(4))) float*)(wconv + co * C * 9);
            float v = bconv[co];
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int ci = 0; ci < CM; ++ci)
                    if (ci < C) v = fmaf(wp[ci * 9 + k], uu[ci][k], v);
            acc[co] = v;
        }
    }
    const float sigma = a.variance == 1 ? cf.sigma_beta : cf.sigma_tilde;
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        if (co < C) {
            const long long e = (((long long)b * C + co) * S + y) * S + x;
            const float eps = acc[co];
            if (a.eps_out) a.eps_out[e] = eps;
            if (a.x_out) {
                float v;
                if (table) {
                    v = row.a * xin[co] + row.b * eps;                  // affine_step_kernel's order and roundings
                    if (draw && a.noise_mode == 1) v = v + row.c * zin[co];
                    else if (draw && a.noise_mode == 2) v = v + row.c * zn[co];
                } else {
                    v = cf.c1 * (xin[co] - cf.c2 * eps);
                    if (t > 0) {
                        if (a.noise_mode == 1) v = v + sigma * zin[co];
                        else if (a.noise_mode == 2) v = v + sigma * zn[co];
                    }
                }
                a.x_out[e] = v;
            }
        }
    }
}

__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ z, float* __restrict__ out, StepCoef c,
                                 int use_noise, int variance_beta, long long n) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = c.c1 * (x[i] - c.c2 * eps[i]);
    if (use_noise) v = v + (variance_beta ? c.sigma_beta : c.sigma_tilde) * z[i];
    out[i] = v;
}

// The same update driven by the device-resident step state (dd_sample_early_exit): t = st->t_final, coefficients from the
// context's table, z from the Philox generator of the fused step (same counter: pixel, t); advance: one thread hands t - 1
// to the next step.  x [B, C, S, S] in place; one thread per pixel.
__global__ void __launch_bounds__(256) ddpm_step_state_kernel(float* __restrict__ x, const float* __restrict__ eps, StepState* st,
                                                              const StepCoef* __restrict__ coef, int B, int C, int S,
                                                              int noise_mode, int advance, long long pix0) {
#pragma clang fp contract(off)
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long hw = (long long)S * S;
    const int t = st->t_final;
    if (advance && pix == 0) { st->t = t - 1; st->t_model = (float)(t - 1); }   // no block of this kernel reads t / t_model
    if (pix >= (long long)B * hw) return;
    const StepCoef cf = coef[t < 0 ? 0 : (t > 999 ? 999 : t)];
    f32x4 zn = {0.f, 0.f, 0.f, 0.f};
    if (t > 0 && noise_mode == 2) zn = philox_normal4(st->seed, (unsigned long long)(pix + pix0), t);     // pix0: a half-batch chain's first pixel within the whole batch
    const long long b = pix / hw, p = pix - b * hw;
    for (int c = 0; c < C; ++c) {
        const long long e = (b * C + c) * hw + p;
        float v = cf.c1 * (x[e] - cf.c2 * eps[e]);
        if (t > 0 && noise_mode == 2) v = v + cf.sigma_tilde * zn[c];
        x[e] = v;
    }
}

// ee_select_kernel and the update above in one launch (the device-resident early-exit loop: two launches less on every step's serial tail,
// and the selected model output never travels through HBM): per pixel, the image's exit layer from cls (ee_select_kernel's rule), then the
// update on (outputs ++ [eps])[idx] -- the same operations in the same order as the two kernels.
__global__ void __launch_bounds__(256) ee_select_step_kernel(float* __restrict__ x, const float* __restrict__ outs, const float* __restrict__ eps,
                                                             const float* __restrict__ cls, float thr, int depth, int* __restrict__ idx_out,
                                                             int idx_stride, int idx_col0, StepState* st, const StepCoef* __restrict__ coef,
                                                             int B, int C, int S, int noise_mode, int advance, long long pix0) {
#pragma clang fp contract(off)
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long hw = (long long)S * S;
    const int t = st->t_final;
    if (advance && pix == 0) { st->t = t - 1; st->t_model = (float)(t - 1); }   // no block of this kernel reads t / t_model
    if (pix >= (long long)B * hw) return;
    const StepCoef cf = coef[t < 0 ? 0 : (t > 999 ? 999 : t)];
    f32x4 zn = {0.f, 0.f, 0.f, 0.f};
    if (t > 0 && noise_mode == 2) zn = philox_normal4(st->seed, (unsigned long long)(pix + pix0), t);
    const long long b = pix / hw, p = pix - b * hw;
    int idx = -1;
    for (int k = 0; k < depth && idx < 0; ++k)
        if (cls[(long long)k * B + b] <= thr) idx = k;
    if (idx < 0) idx = (0.0f <= thr) ? depth : 0;
    if (idx_out && p == 0) idx_out[(long long)t * idx_stride + idx_col0 + b] = idx;     // row t of indices_by_timestep (eesampler.py:71)
    const float* src = idx == depth ? eps : outs + (long long)idx * B * C * hw;
    for (int c = 0; c < C; ++c) {
        const long long e = (b * C + c) * hw + p;
        float v = cf.c1 * (x[e] - cf.c2 * src[e]);
        if (t > 0 && noise_mode == 2) v = v + cf.sigma_tilde * zn[c];
        x[e] = v;
    }
}

// out = a*x + b*m + c*z, each product rounded (no FMA contraction): the common form of the reference's
// predict_original / predict_previous post-processing (sampler.py:59-79) and of a DDIM step (:112-120).
__global__ void affine_step_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                   const float* __restrict__ z, float* __restrict__ out, float a, float b, float c,
                                   long long n) {
#pragma clang fp contract(off)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = a * x[i] + b * m[i];
    if (z) v = v + c * z[i];
    out[i] = v;
}

__global__ void set_state_kernel(StepState* st, int t, unsigned long long seed) {
    st->t = t;
    st->t_final = t;
    st->t_model = (float)t;
    st->seed = seed;
}
__global__ void set_state_table_kernel(StepState* st, const AffineRow* atab, unsigned long long seed) {
    st->t = 0;
    st->t_final = 0;
    st->t_model = atab[0].t_model;
    st->seed = seed;
}
__global__ void set_state_float_kernel(StepState* st, float t) {
    st->t = (int)t;
    st->t_final = (int)t;
    st->t_model = t;
}

// reference sampler.py:145-146: samples = rearrange((x + 1) / 2, "b c h w -> b h w c").  One thread per pixel: the NCHW
// reads are coalesced per channel plane, the NHWC writes are C contiguous floats per thread.
__global__ void __launch_bounds__(256) to_images_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C, int S) {
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long hw = (long long)S * S;
    if (pix >= (long long)B * hw) return;
    const long long b = pix / hw, p = pix - b * hw;
    for (int c = 0; c < C; ++c) out[pix * C + c] = (x[(b * C + c) * hw + p] + 1.0f) / 2.0f;
}

// ---- early-exit baseline (reference models/early_exit.py, eesampler.py) --------------------------------------
// MLPProbe (early_exit.py:31-37): u[b] = mean over the L tokens of sigmoid(x[b,l,:] . w + bias).  One workgroup per
// image, one wave per token row (coalesced 256 B segments), fixed-order reductions (deterministic per image).
// probe row pi = t * t_mul + add with t = the step's timestep on the device (StepState::t_final): per-timestep probes are
// selected inside the launch, so a captured step replays for every t (matrix[key], reference early_exit.py:219-240)
// Two launches with the arithmetic (and every summation order) of the one-workgroup-per-image kernel they replace, which walked an
// image's 257 rows on four waves one after the other (164 us per layer at B = 128: 62 % of the early-exit loop's overhead):
//   rows:   grid (B, slices): each wave takes rows of its slice, two in flight; row value s_l = sigmoid(w . x_l + b) with the per-lane
//           fma chain over k = lane, lane + 64, ... and the xor-shuffle tree -> srow[b, l]
//   reduce: one wave per image, lane i adds s_l for l = i, i + 64, ... in ascending order, xor-shuffle tree; out[b] = sum / L
__global__ void __launch_bounds__(256) ee_probe_rows_kernel(const float* __restrict__ x, const float* __restrict__ w_base,
                                                            const float* __restrict__ bias_base, float* __restrict__ srow, int L,
                                                            int D, const StepState* __restrict__ st, int t_mul, int add) {
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (L + (int)gridDim.y - 1) / (int)gridDim.y;
    const int l0 = blockIdx.y * per, l1 = l0 + per < L ? l0 + per : L;
    const int pi = (t_mul ? st->t_final * t_mul : 0) + add;
    const float* w = w_base + (long long)pi * D;
    const float bv = bias_base[pi];
    for (int l = l0 + wave; l < l1; l += 8) {
        const int la = l, lb = l + 4 < l1 ? l + 4 : l;              // two rows in flight (the second repeats the first past the end)
        const float* xa = x + ((long long)b * L + la) * D;
        const float* xb = x + ((long long)b * L + lb) * D;
        float da = 0.f, db = 0.f;
        for (int k = lane; k < D; k += 64) {
            const float wk = w[k];
            da = fmaf(xa[k], wk, da);
            db = fmaf(xb[k], wk, db);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { da += __shfl_xor(da, o); db += __shfl_xor(db, o); }
        if (lane == 0) {
            srow[(long long)b * L + la] = 1.0f / (1.0f + expf(-(da + bv)));
            if (lb != la) srow[(long long)b * L + lb] = 1.0f / (1.0f + expf(-(db + bv)));
        }
    }
}
// One wave per (image, layer) row of L sigmoids: lane i adds s_l for l = i, i + 64, ... in ascending order (coalesced 256-byte reads), then the
// xor-shuffle tree; out = sum / L.  (Four threads per row walked it with 4-byte strided reads: 36 us for the 13 x 64 rows of one early-exit step
// on 13 workgroups, on every step's critical path.)
__global__ void __launch_bounds__(256) ee_probe_reduce_kernel(const float* __restrict__ srow, float* __restrict__ out, int B, int L) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float acc = 0.f;
    for (int l = lane; l < L; l += 64) acc += srow[(long long)b * L + l];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[b] = acc / (float)L;
}

// AttentionProbe (early_exit.py:40-80): one learned query attends over the tokens after the first, then Linear -> SiLU ->
// Linear(D, 1).  With u = Wk^T q / sqrt(D) folded on the host: p = softmax_l(u . x_l), xbar = sum_l p_l x_l, o = Wv xbar + bv,
// out = w2 . silu(W0 o + b0) + b2.  One workgroup per image, fp32, fixed-order reductions; LDS: L scores + 3 D-vectors.
__global__ void __launch_bounds__(256) ee_attn_probe_kernel(const float* __restrict__ x, const AttnProbeW w, float* __restrict__ out,
                                                            int L, int D) {
    extern __shared__ float ap_lds[];
    float* sc = ap_lds;                 // [L] scores -> probabilities (entry 0 unused: the first token is skipped)
    float* xbar = ap_lds + L;           // [D]
    float* ov = xbar + D;               // [D]
    float* hc = ov + D;                 // [D]
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* xb = x + (long long)b * L * D;
    for (int l = 1 + wave; l < L; l += 4) {
        const float* xr = xb + (long long)l * D;
        float d = 0.f;
        for (int k = lane; k < D; k += 64) d = fmaf(xr[k], w.u[k], d);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) sc[l] = d;
    }
    __syncthreads();
    auto block_reduce = [&](float v, bool is_max) -> float {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const float t = __shfl_xor(v, o); v = is_max ? fmaxf(v, t) : v + t; }
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
    };
    float mx = -3.0e38f;
    for (int l = 1 + tid; l < L; l += 256) mx = fmaxf(mx, sc[l]);
    mx = block_reduce(mx, true);
    float sum = 0.f;
    for (int l = 1 + tid; l < L; l += 256) { const float e = expf(sc[l] - mx); sc[l] = e; sum += e; }
    sum = block_reduce(sum, false);
    const float inv = 1.0f / sum;
    for (int k = tid; k < D; k += 256) {
        float a = 0.f;
        for (int l = 1; l < L; ++l) a = fmaf(sc[l], xb[(long long)l * D + k], a);
        xbar[k] = a * inv;
    }
    __syncthreads();
    for (int j = tid; j < D; j += 256) {
        float a = w.bv[j];
        for (int k = 0; k < D; ++k) a = fmaf(w.wvt[(long long)k * D + j], xbar[k], a);
        ov[j] = a;
    }
    __syncthreads();
    float part = 0.f;
    for (int j = tid; j < D; j += 256) {
        float a = w.b0[j];
        for (int k = 0; k < D; ++k) a = fmaf(w.w0t[(long long)k * D + j], ov[k], a);
        hc[j] = a / (1.0f + expf(-a));                               // SiLU
        part = fmaf(w.w2[j], hc[j], part);
    }
    part = block_reduce(part, false);
    if (tid == 0) out[b] = part + w.b2[0];
}

// eesampler.py:61-67: idx[b] = first layer i in [0, depth] with c[i][b] <= threshold, where c[depth][b] = 0 closes the
// list (torch.argmax of an all-False column is 0); model_output[b] = (outputs ++ [eps])[idx[b]][b].
__global__ void ee_select_kernel(const float* __restrict__ outs, const float* __restrict__ eps, const float* __restrict__ cls,
                                 float thr, int depth, int B, long long chw, float* __restrict__ mo, int* __restrict__ idx_out,
                                 const StepState* __restrict__ st, int idx_stride, int idx_col0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * chw) return;
    // row t of indices_by_timestep (eesampler.py:71); a half-batch chain writes its columns [idx_col0, idx_col0 + B) of the whole batch's row
    if (st && idx_out) idx_out += (long long)st->t_final * idx_stride + idx_col0;
    const int b = (int)(i / chw);
    int idx = -1;
    for (int k = 0; k < depth && idx < 0; ++k)
        if (cls[(long long)k * B + b] <= thr) idx = k;
    if (idx < 0) idx = (0.0f <= thr) ? depth : 0;
    mo[i] = idx == depth ? eps[i] : outs[(long long)idx * B * chw + i];
    if (idx_out && i == (long long)b * chw) idx_out[b] = idx;
}

// eesampler.py:70: per-layer mean over the batch of the predicted errors (logging)
// (scale = 1 / B: the mean; scale = 1: the plain sum -- a half-batch chain's share, ee_mean_combine_kernel divides)
__global__ void __launch_bounds__(64) ee_batch_mean_kernel(const float* __restrict__ cls, float* __restrict__ err, int B,
                                                           const StepState* __restrict__ st, float scale) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (st) err += (long long)st->t_final * gridDim.x;            // row t of error_prediction_by_timestep (eesampler.py:70)
    float a = 0.f;
    for (int b = lane; b < B; b += 64) a += cls[(long long)k * B + b];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) err[k] = scale == 1.0f ? a : a / (float)B;
}

// rows [t_lo, t_hi] of error_prediction_by_timestep from the two chains' per-step sums, in a fixed order: (chain 0 + chain 1) / B
__global__ void ee_mean_combine_kernel(const float* __restrict__ s0, const float* __restrict__ s1, float* __restrict__ err, int depth, int t_lo, int t_hi, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = (t_hi - t_lo + 1) * depth;
    if (i >= n) return;
    const long long e = (long long)t_lo * depth + i;
    err[e] = (s0[e] + s1[e]) / (float)B;
}

}  // namespace

bool embed_mfma_ok(const EmbedArgs& a, size_t& lds) {
    const int pd = a.C * a.P * a.P;
    lds = (size_t)(a.D / 32) * 2 * ((pd / 4 + 3) / 4) * 1024;
    return a.S == 16 * a.P && a.D % 32 == 0 && pd % 4 == 0 && lds <= 156 * 1024 && a.L == a.extras + 256 &&
           ((a.P == 4 && a.C == 3) || (a.P == 2 && a.C == 3) || (a.P == 2 && a.C == 4));
}

// the MFMA kernel's variant that also writes the first block's norm1 in fragment order: the CelebA shape (patch 4, 3 channels, embed_dim 512)
bool embed_ln_supported(const EmbedArgs& a) {
    size_t lds = 0;
    return !a.generic && embed_mfma_ok(a, lds) && a.P == 4 && a.C == 3 && a.D == 512;
}

hipError_t launch_embed(const EmbedArgs& a, hipStream_t s) {
    size_t mlds = 0;
    if (!a.generic && embed_mfma_ok(a, mlds)) {
        dim3 grid((unsigned)((a.B * 16 + 7) / 8));
        if (a.ln_frag) {      // + the first block's norm1 (embed_ln_supported)
            if (!embed_ln_supported(a) || !a.ln_g || !a.ln_b) return hipErrorInvalidValue;
            hipLaunchKernelGGL((embed_mfma_kernel<4, 3, 512>), grid, dim3(512), mlds, s, a);
            return hipGetLastError();
        }
        // few images: deal the column pairs over grid.y until the launch has about one workgroup per CU (an even share of pairs each)
        const int np = a.D / 32;
        while (grid.x * grid.y * 2 <= 256 && np % (grid.y * 4) == 0) grid.y *= 2;
        if (a.P == 4) hipLaunchKernelGGL((embed_mfma_kernel<4, 3>), grid, dim3(512), mlds, s, a);
        else if (a.C == 3) hipLaunchKernelGGL((embed_mfma_kernel<2, 3>), grid, dim3(512), mlds, s, a);
        else hipLaunchKernelGGL((embed_mfma_kernel<2, 4>), grid, dim3(512), mlds, s, a);
        return hipGetLastError();
    }
    const int per_img = (a.L + kEmbedTok - 1) / kEmbedTok;
    const long long pad_rows = (long long)a.Mp - (long long)a.B * a.L;
    const int pad_blocks = (int)((pad_rows + kEmbedTok - 1) / kEmbedTok);
    hipLaunchKernelGGL(embed_kernel, dim3(a.B * per_img + pad_blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_time_mlp(const TimeMlpArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(time_mlp_kernel, dim3(a.B), dim3(256), (size_t)5 * a.D * sizeof(float), s, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, T* out, int rows,
                            int D, hipStream_t s) {
    if (D % 64 || D > 64 * kLnMaxPerLane) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel<T>, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, out, rows, D, (T*)nullptr, 0, 0);
    return hipGetLastError();
}
// the patch rows in MFMA fragment order into `frag` (see layernorm_kernel), the extra-token rows row-major into `out`
hipError_t launch_layernorm_frag(const float* x, const float* gamma, const float* beta, bf16_t* out, bf16_t* frag, int rows, int D,
                                 int tok_l, int tok_e, hipStream_t s) {
    if (D % 256 || D > 64 * kLnMaxPerLane || !frag || tok_l <= tok_e || (tok_l - tok_e) % 32 || rows % tok_l) return hipErrorInvalidValue;
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, s, x, gamma, beta, out, rows, D, frag, tok_l, tok_e);
    return hipGetLastError();
}
hipError_t launch_reduce_ln(const ReduceLnArgs& a, int D, hipStream_t s) {
    if (D % 256 || D > 1024 || a.rows < 1 || a.splits < 1 || !a.x || !a.partial || !a.bias || (a.ln_g && (!a.ln_b || !a.h)) ||
        (a.frag && (a.tok_l <= a.tok_e || (a.tok_l - a.tok_e) % 32 || a.rows % a.tok_l)))
        return hipErrorInvalidValue;
    const dim3 grid((a.rows + 3) / 4);
    switch (D / 256) {
        case 1: hipLaunchKernelGGL(reduce_ln_kernel<1>, grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(reduce_ln_kernel<2>, grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(reduce_ln_kernel<3>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(reduce_ln_kernel<4>, grid, dim3(256), 0, s, a); break;
    }
    return hipGetLastError();
}
template hipError_t launch_layernorm<bf16_t>(const float*, const float*, const float*, bf16_t*, int, int, hipStream_t);
template hipError_t launch_layernorm<float>(const float*, const float*, const float*, float*, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------
// Output head, first half (reference models/uvit.py:377-378): dec = decoder_pred(norm(x)) in one launch, exact fp32
// (v_mfma_f32_16x16x4_f32 == an fmaf chain), so eps never sees a bf16 rounding and the normalised rows never exist in HBM.
// The host folds the affine part of the LayerNorm into the Linear once (finalize), and the kernel the normalisation itself
// out of the product, so that the MFMAs run on the rows AS THEY ARRIVE instead of behind the last load and the statistics:
//     dec[m] = sum_k (W[m,k] gamma[k]) (x[k] - mean) rstd + (b[m] + sum_k W[m,k] beta[k])
//            = rstd * ( Wg[m,:] . d  -  mean_d * wsum[m] )  +  c[m],        d = x - x[0],  mean_d = mean(d),  wsum[m] = sum_k Wg[m,k]
// (the shift by the row's own first element keeps |mean_d| within the row's spread whatever offset the row carries, so the
// subtraction loses nothing a LayerNorm of the same row would keep; rstd from the two-pass variance of d, as layernorm_kernel).
// Workgroup = 128 rows, wave = 16 rows held in registers in MFMA B-operand order (lane: row l & 15, k-quad l >> 4);
// Wg is parked in LDS in A-operand order once per workgroup.  A wave's row loads are all in flight before it waits for its
// share of Wg; the MFMAs of k-block j wait for load j only (counted vmcnt), the statistics follow on the registers.
// HBM: the fp32 rows once (the launch's roof), dec once.
// ------------------------------------------------------------------------------------------
// 16-byte global load the compiler does not count, and the wait that makes its destination usable ("+v": every use is ordered behind it)
// HAND = false: an ordinary load / nothing -- hipcc counts and waits itself (D >= 768: the row quads + the Wg share are more than the 63 operations
// the counter holds, and more registers than the 256 VGPRs a wave has: hipcc parks quads in AGPRs, which must not happen to a register whose load
// is still in flight)
template <int OFF, bool HAND>
__device__ __forceinline__ void asm_load_quad(f32x4& dst, const float* p) {
    if constexpr (HAND) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "i"(OFF));
    else dst = *reinterpret_cast<const f32x4*>(p + OFF / 4);
}
template <int N, bool HAND>
__device__ __forceinline__ void landed_at_vmcnt(f32x4& v) {
    static_assert(!HAND || N < 64, "a 6-bit counter");
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N));
}

// NW waves per workgroup: 8 (two per SIMD: D <= 512, D / 4 row registers per lane), 4 at D = 768 / 1024 (192 / 256 row registers: one wave per SIMD).
// PROBE (early-exit heads, reference models/early_exit.py:31-37): the launch also writes the MLP probe's per-token value
//     srow[row] = sigmoid(x[row,:] . w + b)        (w, b: probe row t * t_mul + add, t read from the step state as in ee_probe_rows_kernel)
// for the rows it decodes -- the quads are in registers anyway: 4 VALU fmas per quad under the MFMAs instead of a second pass over the
// rows in HBM (13 launches per step) -- and, patch-rows-only launches, for the extra-token rows in a short pass of their own behind the units.
// SPLIT (the early-exit heads of the bf16 engine; the model's final head and the fp32 engine keep the exact product): Wg . d as a split-bf16
// product on v_mfma_f32_16x16x32_bf16 -- Wg = Wh + Wl and d = dh + dl, each half a bf16 (the low half = the bf16 of the remainder), and
// Wh.dh + Wh.dl + Wl.dh accumulated in fp32; the dropped Wl.dl term and the low halves' own rounding are 2^-16 .. 2^-17 of a product, two
// orders below the bf16 roundings of the backbone that feeds these heads -- 9 MFMAs of 16 cycles per 32 k instead of 24 of 32 cycles (a fifth
// of the matrix-pipe time, and of its energy).  Lane (row l & 15, k-group l >> 4) holds k = 32 jj + 8 (l >> 4) .. + 7 as TWO quads per jj;
// a.wg is the host-packed image [jj][ct][hi, lo][64 lanes] x 8 bf16 in A-operand order (capi.hip: pack_head_split).
__device__ __forceinline__ unsigned head_pack2_bf16(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (round to nearest even)
    typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
    typedef float f32v2 __attribute__((ext_vector_type(2)));
    const f32v2 q = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16v2));
}

template <int D, int NT, int NW = 8, bool PROBE = false, bool SPLIT = false>
__global__ void __launch_bounds__(NW * 64) head_dec_kernel(const HeadDecArgs a) {
    constexpr int J = D / 16;
    // quad jq of a lane: byte offset from the lane's base (row, k-group) and its index in the probe row's quads
    constexpr auto quad_off = [](int jq) constexpr { return SPLIT ? 128 * (jq / 2) + 16 * (jq % 2) : 64 * jq; };
    constexpr int KG = SPLIT ? 8 : 4;      // floats between the k-groups' bases
    constexpr bool HAND = J + J * NT / NW + (PROBE ? 1 : 0) <= 52;     // D <= 512: J row quads + the Wg share <= 63 operations in flight, all of them in VGPRs (<= 208 of 256)
    extern __shared__ __attribute__((aligned(16))) char head_lds[];
    f32x4* wl = reinterpret_cast<f32x4*>(head_lds);                 // [J][NT][64]
    f32x4* pwl = wl + J * NT * 64;                                  // PROBE: [D / 4] the probe row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    static_assert(!PROBE || (D % 256 == 0 && D / 256 <= NW), "the probe row is staged by D / 256 waves");
    int pi = 0;
    if constexpr (PROBE) pi = (a.t_mul ? a.st->t_final * a.t_mul : 0) + a.add;
    // 16-row units dealt to the waves of the whole grid, unit u -> wave (u / grid) % NW of workgroup u % grid: with
    // M = B (256 + extras) rows the few units beyond one per wave land on different CUs (128-row tiles per workgroup
    // left one workgroup for a second round of the whole launch).  Nothing below the barrier synchronises across waves.
    // tok_l > 0: only the patch rows (tokens l >= tok_e of every tok_l-row image; 16 | tok_l - tok_e) are decoded -- the extra tokens'
    // rows of dec are never read (unpatchify takes the patch tokens, reference models/uvit.py:379-381) and at B = 128 those 128 rows
    // were 8 units more than one per wave: a second round of the whole launch on 8 CUs
    const int upi = a.tok_l > 0 ? (a.tok_l - a.tok_e) / 16 : 0;                       // units per image
    const int units = a.tok_l > 0 ? (a.M / a.tok_l) * upi : (a.M + 15) / 16, stride = NW * (int)gridDim.x;
    auto row_of = [&](int u) -> long long {
        return a.tok_l > 0 ? (long long)(u / upi) * a.tok_l + a.tok_e + (u % upi) * 16 + n : (long long)u * 16 + n;
    };
    // The row quads and the Wg fragments travel as asm loads: hipcc does not count those, so every wait below is written by hand
    // (vector memory returns in order: "at most N outstanding" = everything but the youngest N has landed; loads the compiler
    // issues in between only make a hand-written wait stricter).  A compiler-counted version waited for ALL row loads in front
    // of the first MFMA (its counters merge conservatively around the unit loop).
    f32x4 xv[J];
    auto request = [&](int u) {     // this lane's quads of its row of unit u (rows past M: the last row, never stored)
        const long long row = row_of(u);
        const float* xr = a.x + (row < a.M ? row : (long long)a.M - 1) * D + KG * q;
        [&]<int... JJ>(std::integer_sequence<int, JJ...>) {
            (asm_load_quad<quad_off(JJ), HAND>(xv[JJ], xr), ...);
        }(std::make_integer_sequence<int, J>{});
    };
    // Wg fragments of this wave (L2-resident) -> LDS in A-operand order; requested FIRST, the first unit's rows right behind
    // them: the rows are in flight while the workgroup parks Wg and meets at the barrier
    constexpr int WI = J * NT / NW;
    static_assert(J * NT % NW == 0, "one equal share of Wg fragments per wave");
    static_assert(quad_off(J - 1) < 4096, "the row quads' immediate offsets");
    static_assert(!SPLIT || (HAND && J % 2 == 0), "the split product is built for the hand-counted widths");
    static_assert(!HAND || J + WI <= 63, "the vector-memory counter holds 63 operations");
    int u = wave * (int)gridDim.x + (int)blockIdx.x;
    {
        f32x4 wv[WI];
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int item = wave + NW * i, j = item / NT, ct = item - j * NT, m = 16 * ct + n;
            const float* wp = SPLIT ? a.wg + (long long)item * 256 + lane * 4      // (the packed image: item = (jj NT + ct) 2 + half, 1 KB each)
                                    : a.wg + (long long)(m < a.pd ? m : 0) * D + 16 * j + 4 * q;   // (rows m >= pd feed output columns that are never stored)
            asm_load_quad<0, HAND>(wv[i], wp);
        }
        // (every wave requests a quad of the probe row, D / 256 of them park theirs: an asm load under a branch hands hipcc a phi
        // to copy -- out of a register whose load is still in flight)
        f32x4 pwq;
        if constexpr (PROBE) asm_load_quad<0, HAND>(pwq, a.pw_base + (long long)pi * D + (wave % (D / 256)) * 256 + lane * 4);
        if (u < units) request(u);
        else if constexpr (HAND) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (a wave without a unit: nothing younger than its fragments)
        [&]<int... II>(std::integer_sequence<int, II...>) {                     // the fragments have landed when at most the J row loads are outstanding
            (landed_at_vmcnt<J, HAND>(wv[II]), ...);
        }(std::make_integer_sequence<int, WI>{});
#pragma unroll
        for (int i = 0; i < WI; ++i) wl[(wave + NW * i) * 64 + lane] = wv[i];
        if constexpr (PROBE) {
            landed_at_vmcnt<J, HAND>(pwq);            // (requested in front of the row quads)
            if (wave < D / 256) pwl[wave * 64 + lane] = pwq;
        }
    }
    __syncthreads();
    while (u < units) {
        const long long row = row_of(u);
        const bool ok = row < a.M;
        typedef const __attribute__((address_space(3))) f32x4* lds_quad_ptr;   // (kept in the LDS address space: behind the opaque asm a generic
        lds_quad_ptr wlp = (lds_quad_ptr)wl + lane;                           //  pointer turns every fragment read into a flat load, which counts in vmcnt too)
        asm volatile("" : "+v"(wlp));   // opaque per unit: the LDS fragment reads are loop-invariant and hipcc would hoist all of them (spills)
        // the shift: the row's first element (held by the lane of k-quad 0)
        landed_at_vmcnt<J - 1, HAND>(xv[0]);
        const float x0 = __shfl(xv[0][0], n);
        f32x4 acc[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        lds_quad_ptr pwp = (lds_quad_ptr)pwl + (SPLIT ? 2 * q : q);
        asm volatile("" : "+v"(pwp));
        f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
        [&]<int... JJ>(std::integer_sequence<int, JJ...>) {      // fully unrolled: xv[] must stay in registers
            ([&] {
                constexpr int j = JJ;
                landed_at_vmcnt<J - 1 - j, HAND>(xv[j]);      // quad j has landed
                if constexpr (PROBE) {
                    const f32x4 pw4 = pwp[SPLIT ? 8 * (j / 2) + (j % 2) : 4 * j];      // (pwp = the probe row's quads + q (2 q with SPLIT))
#pragma unroll
                    for (int e = 0; e < 4; ++e) pacc[e] = fmaf(xv[j][e], pw4[e], pacc[e]);
                    asm volatile("" : "+v"(pacc));      // (here, not after the loop: hipcc parked every quad and probe weight in scratch to run the 128 fmas at the end)
                }
                xv[j] = xv[j] - x0;
                if constexpr (!SPLIT) {
                    f32x4 w[NT];
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) w[ct] = wlp[(j * NT + ct) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int ct = 0; ct < NT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ct][e], xv[j][e], acc[ct], 0, 0, 0);
                } else if constexpr (j % 2 == 1) {      // both quads of k-block jj = j / 2 are here: split, then 3 NT MFMAs
                    constexpr int jj = j / 2;
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 hi, lo;
#pragma unroll
                    for (int p2 = 0; p2 < 4; ++p2) {
                        const float d0 = p2 < 2 ? xv[j - 1][2 * p2] : xv[j][2 * p2 - 4], d1 = p2 < 2 ? xv[j - 1][2 * p2 + 1] : xv[j][2 * p2 - 3];
                        const unsigned h = head_pack2_bf16(d0, d1);
                        hi[p2] = h;
                        lo[p2] = head_pack2_bf16(d0 - __builtin_bit_cast(float, h << 16), d1 - __builtin_bit_cast(float, h & 0xffff0000u));
                    }
                    const bf16x8 dh = __builtin_bit_cast(bf16x8, hi), dl = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, wlp[((jj * NT + ct) * 2) * 64]), wlo = __builtin_bit_cast(bf16x8, wlp[((jj * NT + ct) * 2 + 1) * 64]);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dh, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, dl, acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, dh, acc[ct], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);     // (hipcc would sink this block's MFMAs below the later waits)
            }(), ...);
        }(std::make_integer_sequence<int, J>{});
        // c and wsum of this lane's output columns are fetched HERE, per unit, through a pointer hipcc cannot see through: loaded in
        // front of the unit loop they are compiler-counted loads older than the row quads -- hipcc waited for them (vmcnt(0), i.e. for
        // every row quad) before the first MFMA
        const float* cw = a.c;
        asm volatile("" : "+s"(cw));
        f32x4 c4[NT], ws4[NT];
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            const int col = 16 * ct + 4 * q < a.pd ? 16 * ct + 4 * q : 0;
            c4[ct] = *reinterpret_cast<const f32x4*>(cw + col);
            ws4[ct] = *reinterpret_cast<const f32x4*>(cw + a.pd + col);
        }
        // statistics of d on the registers (two passes), under the other wave's MFMAs
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < J; ++j) s4 += xv[j];
        float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float mean = sum / (float)D;
        f32x4 q4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f32x4 d = xv[j] - mean;
            q4 += d * d;
        }
        float sq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq / (float)D + 1e-5f);
        // (used here, outside the branch: a wait for c / wsum that exists on the storing path only leaves them "in flight" on the other
        // one, and hipcc then waits vmcnt(0) -- for every row quad -- where the next unit first overwrites their registers)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) asm volatile("" : "+v"(c4[ct]), "+v"(ws4[ct]));
        if (ok) {      // lane holds dec[row][16 ct + 4 q + i], i < 4 (pd % 4 == 0: a quad is inside or outside)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                const int col = 16 * ct + 4 * q;
                if (col < a.pd) *reinterpret_cast<f32x4*>(a.dec + row * a.pd + col) = (acc[ct] - mean * ws4[ct]) * rstd + c4[ct];
            }
        }
        if constexpr (PROBE) {
            float pd = (pacc[0] + pacc[1]) + (pacc[2] + pacc[3]);
            pd += __shfl_xor(pd, 16);
            pd += __shfl_xor(pd, 32);
            if (ok && q == 0) a.srow[row] = 1.0f / (1.0f + expf(-(pd + a.pb_base[pi])));
        }
        const int un = u + stride;
        if (un < units) request(un);
        u = un;
    }
    if constexpr (PROBE) {
        // patch-rows-only launch: the probe's values for the extra-token rows (tok_e per image), 16 rows per wave 0 of the first workgroups
        if (a.tok_l > 0 && a.tok_e > 0 && wave == 0) {
            const int nex = (a.M / a.tok_l) * a.tok_e;
            for (int eu = (int)blockIdx.x; eu * 16 < nex; eu += (int)gridDim.x) {
                const int idx = eu * 16 + n, ii = idx < nex ? idx : nex - 1;
                const long long row = (long long)(ii / a.tok_e) * a.tok_l + ii % a.tok_e;
                const float* xr = a.x + row * D + 4 * q;
                f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int j = 0; j < J; ++j) {
                    const f32x4 xq = *reinterpret_cast<const f32x4*>(xr + 16 * j), pw4 = pwl[4 * j + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) pacc[e] = fmaf(xq[e], pw4[e], pacc[e]);
                }
                float pd = (pacc[0] + pacc[1]) + (pacc[2] + pacc[3]);
                pd += __shfl_xor(pd, 16);
                pd += __shfl_xor(pd, 32);
                if (idx < nex && q == 0) a.srow[row] = 1.0f / (1.0f + expf(-(pd + a.pb_base[pi])));
            }
        }
    }
}

bool head_dec_supported(int D, int pd) {
    if (pd < 1 || pd > 64 || pd % 4) return false;
    const int nt = (pd + 15) / 16;
    return D == 256 || D == 512 || (D == 768 && nt <= 3) || (D == 1024 && nt <= 2);     // (D / 16) nt KB of LDS for Wg
}

bool head_dec_probe_supported(int D) { return D == 256 || D == 512; }

void pack_head_split(int D, int pd, const float* wg, unsigned short (*f2bf)(float), unsigned short* img, float* wsum) {
    const int nt = (pd + 15) / 16, J2 = D / 32;
    auto bf2f = [](unsigned short b) { const unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; };
    std::vector<double> sums(pd, 0.0);
    for (int jj = 0; jj < J2; ++jj)
        for (int ct = 0; ct < nt; ++ct)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = 16 * ct + (lane & 15), k0 = 32 * jj + 8 * (lane >> 4);
                unsigned short* hi = img + ((size_t)((jj * nt + ct) * 2) * 64 + lane) * 8;
                unsigned short* lo = hi + 64 * 8;
                for (int e = 0; e < 8; ++e) {
                    const float w = m < pd ? wg[(size_t)m * D + k0 + e] : 0.f;       // (rows m >= pd feed output columns that are never stored)
                    hi[e] = f2bf(w);
                    lo[e] = f2bf(w - bf2f(hi[e]));
                    if (m < pd) sums[m] += (double)bf2f(hi[e]) + (double)bf2f(lo[e]);
                }
            }
    for (int m = 0; m < pd; ++m) wsum[m] = (float)sums[m];
}

hipError_t launch_head_dec(const HeadDecArgs& a, int D, int num_cus, hipStream_t s) {
    if (!head_dec_supported(D, a.pd) || a.M < 1) return hipErrorInvalidValue;
    if (a.tok_l > 0 && (a.tok_e < 0 || a.tok_e >= a.tok_l || (a.tok_l - a.tok_e) % 16 || a.M % a.tok_l)) return hipErrorInvalidValue;
    if (a.srow && (!head_dec_probe_supported(D) || !a.pw_base || !a.pb_base || (a.t_mul && !a.st))) return hipErrorInvalidValue;
    if (a.split && !head_dec_probe_supported(D)) return hipErrorInvalidValue;     // (the same widths: 256 / 512)
    const int nt = (a.pd + 15) / 16;
    const int wgs = (a.M + 127) / 128;
    const dim3 grid((unsigned)(wgs < num_cus ? wgs : num_cus));     // one workgroup per CU (LDS), 16-row units dealt inside
    const size_t lds = (size_t)(D / 16) * nt * 1024 + (a.srow ? (size_t)D * 4 : 0);
#define DD_HEAD(DV, NV)                                                                                  \
    do {                                                                                                 \
        if (a.split && a.srow) hipLaunchKernelGGL((head_dec_kernel<DV, NV, 8, true, true>), grid, dim3(512), lds, s, a);   \
        else if (a.split) hipLaunchKernelGGL((head_dec_kernel<DV, NV, 8, false, true>), grid, dim3(512), lds, s, a);       \
        else if (a.srow) hipLaunchKernelGGL((head_dec_kernel<DV, NV, 8, true>), grid, dim3(512), lds, s, a);  \
        else hipLaunchKernelGGL((head_dec_kernel<DV, NV>), grid, dim3(512), lds, s, a);                  \
    } while (0)
    if (D == 1024) {
        if (nt == 1) hipLaunchKernelGGL((head_dec_kernel<1024, 1, 4>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((head_dec_kernel<1024, 2, 4>), grid, dim3(256), lds, s, a);
    } else if (D == 768) {
#define DD_HEAD4(NV) hipLaunchKernelGGL((head_dec_kernel<768, NV, 4>), grid, dim3(256), lds, s, a)
        if (nt == 1) DD_HEAD4(1); else if (nt == 2) DD_HEAD4(2); else if (nt == 3) DD_HEAD4(3); else return hipErrorInvalidValue;
#undef DD_HEAD4
    }
    else if (D == 512) { if (nt == 1) DD_HEAD(512, 1); else if (nt == 2) DD_HEAD(512, 2); else if (nt == 3) DD_HEAD(512, 3); else DD_HEAD(512, 4); }
    else { if (nt == 1) DD_HEAD(256, 1); else if (nt == 2) DD_HEAD(256, 2); else if (nt == 3) DD_HEAD(256, 3); else DD_HEAD(256, 4); }
#undef DD_HEAD
    return hipGetLastError();
}

hipError_t init_rowops_kernels() {
    hipError_t e = hipSuccess;
#define DD_HEAD_ATTR(DV, NV)                                                                                       \
    if (e == hipSuccess)                                                                                           \
        e = hipFuncSetAttribute((const void*)head_dec_kernel<DV, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (DV / 16) * NV * 1024); \
    if (e == hipSuccess)                                                                                           \
        e = hipFuncSetAttribute((const void*)head_dec_kernel<DV, NV, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (DV / 16) * NV * 1024 + DV * 4); \
    if (e == hipSuccess)                                                                                           \
        e = hipFuncSetAttribute((const void*)head_dec_kernel<DV, NV, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (DV / 16) * NV * 1024 + DV * 4); \
    if (e == hipSuccess)                                                                                           \
        e = hipFuncSetAttribute((const void*)head_dec_kernel<DV, NV, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (DV / 16) * NV * 1024);
    DD_HEAD_ATTR(512, 1) DD_HEAD_ATTR(512, 2) DD_HEAD_ATTR(512, 3) DD_HEAD_ATTR(512, 4)
    DD_HEAD_ATTR(256, 1) DD_HEAD_ATTR(256, 2) DD_HEAD_ATTR(256, 3) DD_HEAD_ATTR(256, 4)
#undef DD_HEAD_ATTR
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_dec_kernel<768, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 48 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_dec_kernel<768, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_dec_kernel<768, 3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_dec_kernel<1024, 1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)head_dec_kernel<1024, 2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_mfma_kernel<4, 3, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_mfma_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_mfma_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)embed_mfma_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    return e;
}

namespace {
// deterministic pseudo-random fill for the GEMM development harness (uniform [-1, 1))
template <typename T>
__global__ void fill_random_kernel(T* p, long long n, unsigned seed, float scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    p[i] = Elem<T>::from_f32(((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
}
}  // namespace
template <typename T>
hipError_t launch_fill_random(T* p, long long n, unsigned seed, float scale, hipStream_t s) {
    hipLaunchKernelGGL(fill_random_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, seed, scale);
    return hipGetLastError();
}
template hipError_t launch_fill_random<bf16_t>(bf16_t*, long long, unsigned, float, hipStream_t);
template hipError_t launch_fill_random<float>(float*, long long, unsigned, float, hipStream_t);

hipError_t launch_final(const FinalArgs& a, hipStream_t s) {
    if (a.S >= 16) {
        const int tiles = (a.S + 15) / 16;
        const dim3 grid(a.B * tiles * tiles);
        if (a.C == 3 && a.P == 4) hipLaunchKernelGGL((final_tiled_kernel<3, 4>), grid, dim3(256), 0, s, a);          // CelebA-64, ImageNet-64
        else if (a.C == 3 && a.P == 2) hipLaunchKernelGGL((final_tiled_kernel<3, 2>), grid, dim3(256), 0, s, a);     // CIFAR-10
        else if (a.C == 4 && a.P == 2) hipLaunchKernelGGL((final_tiled_kernel<4, 2>), grid, dim3(256), 0, s, a);     // latent 32 x 32 x 4
        else hipLaunchKernelGGL((final_tiled_kernel<0, 0>), grid, dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const long long npix = (long long)a.B * a.S * a.S;
    hipLaunchKernelGGL(final_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_ddpm_step(const float* x, const float* eps, const float* z, float* out, StepCoef c,
                            int use_noise, long long n, hipStream_t s) {
    // variance selection is folded by the caller into c.sigma_tilde
    hipLaunchKernelGGL(ddpm_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, eps, z, out, c,
                       use_noise, 0, n);
    return hipGetLastError();
}

hipError_t launch_affine_step(const float* x, const float* m, const float* z, float* out, float a, float b, float c,
                              long long n, hipStream_t s) {
    hipLaunchKernelGGL(affine_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, m, z, out, a, b, c, n);
    return hipGetLastError();
}

hipError_t launch_ee_probe(const float* x, const float* w_base, const float* bias_base, float* out, float* srow, int B, int L, int D,
                           const StepState* st, int t_mul, int add, hipStream_t s) {
    hipLaunchKernelGGL(ee_probe_rows_kernel, dim3(B, 8), dim3(256), 0, s, x, w_base, bias_base, srow, L, D, st, t_mul, add);
    if (out) hipLaunchKernelGGL(ee_probe_reduce_kernel, dim3((B + 3) / 4), dim3(256), 0, s, srow, out, B, L);     // (out == null: launch_ee_probe_reduce later, for several layers at once)
    return hipGetLastError();
}
// the reduce half alone, for `rows` (image, layer) rows of L sigmoids each: srow [rows, L] -> out [rows], the per-row arithmetic of launch_ee_probe
hipError_t launch_ee_probe_reduce(const float* srow, float* out, int rows, int L, hipStream_t s) {
    hipLaunchKernelGGL(ee_probe_reduce_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, srow, out, rows, L);
    return hipGetLastError();
}
hipError_t launch_ee_attn_probe(const float* x, const AttnProbeW& w, float* out, int B, int L, int D, hipStream_t s) {
    hipLaunchKernelGGL(ee_attn_probe_kernel, dim3(B), dim3(256), (size_t)(L + 3 * D) * sizeof(float), s, x, w, out, L, D);
    return hipGetLastError();
}
hipError_t launch_ee_select(const float* outs, const float* eps, const float* cls, float thr, int depth, int B, long long chw,
                            float* mo, int* idx, float* err_mean, const StepState* st, hipStream_t s, int idx_stride, int idx_col0, bool sums) {
    const long long n = (long long)B * chw;
    hipLaunchKernelGGL(ee_select_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, outs, eps, cls, thr, depth, B, chw, mo, idx, st,
                       idx_stride > 0 ? idx_stride : B, idx_col0);
    if (err_mean) hipLaunchKernelGGL(ee_batch_mean_kernel, dim3(depth), dim3(64), 0, s, cls, err_mean, B, st, sums ? 1.0f : 0.0f);
    return hipGetLastError();
}
hipError_t launch_ee_select_step(float* x, const float* outs, const float* eps, const float* cls, float thr, int depth, int* idx, float* err_mean,
                                 int idx_stride, int idx_col0, bool sums, StepState* st, const StepCoef* coef, int B, int C, int S,
                                 int noise_mode, int advance, hipStream_t s) {
    const long long npix = (long long)B * S * S;
    if (err_mean) hipLaunchKernelGGL(ee_batch_mean_kernel, dim3(depth), dim3(64), 0, s, cls, err_mean, B, st, sums ? 1.0f : 0.0f);
    hipLaunchKernelGGL(ee_select_step_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, outs, eps, cls, thr, depth, idx,
                       idx_stride > 0 ? idx_stride : B, idx_col0, st, coef, B, C, S, noise_mode, advance, (long long)idx_col0 * S * S);
    return hipGetLastError();
}
hipError_t launch_ee_mean_combine(const float* s0, const float* s1, float* err, int depth, int t_lo, int t_hi, int B, hipStream_t s) {
    const int n = (t_hi - t_lo + 1) * depth;
    hipLaunchKernelGGL(ee_mean_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, s0, s1, err, depth, t_lo, t_hi, B);
    return hipGetLastError();
}
hipError_t launch_ddpm_step_state(float* x, const float* eps, StepState* st, const StepCoef* coef, int B, int C, int S,
                                  int noise_mode, int advance, hipStream_t s, int b0) {
    const long long npix = (long long)B * S * S;
    hipLaunchKernelGGL(ddpm_step_state_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, eps, st, coef, B, C, S,
                       noise_mode, advance, (long long)b0 * S * S);
    return hipGetLastError();
}

hipError_t launch_set_state(StepState* st, int t, unsigned long long seed, hipStream_t s) {
    hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(1), 0, s, st, t, seed);
    return hipGetLastError();
}
hipError_t launch_set_state_table(StepState* st, const AffineRow* atab, unsigned long long seed, hipStream_t s) {
    hipLaunchKernelGGL(set_state_table_kernel, dim3(1), dim3(1), 0, s, st, atab, seed);
    return hipGetLastError();
}
hipError_t launch_set_state_float(StepState* st, float t, hipStream_t s) {
    hipLaunchKernelGGL(set_state_float_kernel, dim3(1), dim3(1), 0, s, st, t);
    return hipGetLastError();
}
hipError_t launch_to_images(const float* x, float* out, int B, int C, int S, hipStream_t s) {
    const long long npix = (long long)B * S * S;
    hipLaunchKernelGGL(to_images_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, x, out, B, C, S);
    return hipGetLastError();
}

}  // namespace dd
