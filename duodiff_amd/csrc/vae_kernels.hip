// Bandwidth kernels of the KL-VAE decoder (reference models/utils/autoencoder.py:320-449): layout
// changes, im2col for the 3x3 convolutions (optionally through a nearest-2x upsample), GroupNorm(32)
// + swish, row softmax of the single-head attention block.  All convolutions themselves run as
// GEMMs on the MFMA kernels of gemm.hip (NHWC activations: a pixel is a GEMM row).
#include "dd_internal.h"

namespace dd {
namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// z [B,4,H,W] fp32 NCHW -> (1/scale) z -> post_quant_conv 1x1 (4->4, autoencoder.py:486-488) -> NHWC fp32 [B*H*W, 4]
__global__ void vae_input_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ b,
                                 float inv_scale, float* __restrict__ out, int B, int HW) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * HW) return;
    const int bi = (int)(i / HW), p = (int)(i % HW);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = z[((long long)bi * 4 + c) * HW + p] * inv_scale;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float a = b[o];
#pragma unroll
        for (int c = 0; c < 4; ++c) a = fmaf(w[o * 4 + c], v[c], a);
        out[i * 4 + o] = a;
    }
}

// NHWC fp32 [B*H*W, ldc] (first C channels) -> NCHW fp32 [B,C,H,W]
__global__ void vae_output_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW, int ldc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * C * HW) return;
    const int p = (int)(i % HW), c = (int)((i / HW) % C), bi = (int)(i / ((long long)HW * C));
    out[i] = in[((long long)bi * HW + p) * ldc + c];
}

// im2col for a 3x3 / pad 1 convolution over NHWC `src` [B, Hs, Ws, C]; when up == 1 the convolution input is the
// nearest-2x upsampled image (autoencoder.py:56-59), gathered on the fly.  Row = output pixel, column order
// (ky, kx, c) (the weights are repacked to match), columns [9C, Kpad) are zero.  8 channels (16 B) per thread.
template <typename T>
__global__ void im2col3x3_kernel(const T* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int C, int up,
                                 int Kpad) {
    constexpr int V = 16 / (int)sizeof(T);
    const int cv = C / V;                       // vectors per pixel (C % V == 0), or 1 for the C = 4 input conv
    const long long per_row = (long long)Kpad / V;
    const long long total = (long long)B * H * W * per_row;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long row = i / per_row;
    const int kc = (int)(i % per_row);
    f32x4 val = {0.f, 0.f, 0.f, 0.f};
    const int tap = kc / cv, c0 = (kc % cv) * V;
    if (tap < 9) {
        const int x = (int)(row % W), y = (int)((row / W) % H), bi = (int)(row / ((long long)W * H));
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const int Hs = up ? H / 2 : H, Ws = up ? W / 2 : W;
            const int ys = up ? yy / 2 : yy, xs = up ? xx / 2 : xx;
            val = *reinterpret_cast<const f32x4*>(src + (((long long)bi * Hs + ys) * Ws + xs) * C + c0);
        }
    }
    *reinterpret_cast<f32x4*>(dst + row * Kpad + (long long)kc * V) = val;
}

// conv_in has C = 4 input channels: K = 36, padded to Kpad; scalar gather
template <typename T>
__global__ void im2col3x3_c4_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int H, int W, int Kpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)B * H * W * Kpad;
    if (i >= total) return;
    const long long row = i / Kpad;
    const int k = (int)(i % Kpad);
    float v = 0.f;
    if (k < 36) {
        const int tap = k / 4, c = k % 4;
        const int x = (int)(row % W), y = (int)((row / W) % H), bi = (int)(row / ((long long)W * H));
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = src[(((long long)bi * H + yy) * W + xx) * 4 + c];
    }
    dst[i] = Elem<T>::from_f32(v);
}

// GroupNorm(32 groups, eps 1e-6) in three deterministic passes over the NHWC fp32 stream:
//   gn_stats    one workgroup per (image, chunk of kGnChunk pixels): per-group (sum, sum of squares) partials, fp32,
//               16-byte loads (a float4 = 4 channels of one group since C/32 >= 4), fixed-order LDS reduction;
//   gn_finalize one thread per (image, group): partials combined in double -> (mean, rstd);
//   gn_apply    y = swish?((x - mean) * rstd * gamma + beta) -> T, 16-byte loads.
constexpr int kGnChunk = 256;
__global__ void __launch_bounds__(256) gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, int HW,
                                                       int C, int chunks) {
    __shared__ float ss[256], sq[256];
    const int bi = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const int c4 = C >> 2;                       // float4 columns per pixel: 32, 64 or 128
    const int rows = 256 / c4;                   // pixels covered per pass
    const int col = threadIdx.x % c4, r0 = threadIdx.x / c4;
    const int p0 = ch * kGnChunk, p1 = min(HW, p0 + kGnChunk);
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + (long long)bi * HW * C) + col;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    for (int p = p0 + r0; p < p1; p += rows) {
        const f32x4 v = xp[(long long)p * c4];
        s += v;
        q += v * v;
    }
    ss[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
    sq[threadIdx.x] = (q[0] + q[1]) + (q[2] + q[3]);
    __syncthreads();
    if (threadIdx.x < 32) {                      // group g owns float4 columns [g*w, (g+1)*w), w = c4/32
        const int w = c4 >> 5;
        float a = 0.f, b2 = 0.f;
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < w; ++k) {
                const int t = r * c4 + threadIdx.x * w + k;
                a += ss[t];
                b2 += sq[t];
            }
        float* o = part + (((long long)bi * chunks + ch) * 32 + threadIdx.x) * 2;
        o[0] = a;
        o[1] = b2;
    }
}

__global__ void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ stat, int n_img_groups, int chunks,
                                   double count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // (image, group)
    if (i >= n_img_groups) return;
    const int bi = i >> 5, g = i & 31;
    double s = 0.0, q = 0.0;
    for (int k = 0; k < chunks; ++k) {
        const float* o = part + (((long long)bi * chunks + k) * 32 + g) * 2;
        s += o[0];
        q += o[1];
    }
    const double mean = s / count;
    double var = q / count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stat[2 * i] = (float)mean;
    stat[2 * i + 1] = (float)(1.0 / sqrt(var + 1e-6));
}

template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ stat,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ out, int HW, int C, int swish, long long total4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index
    if (i >= total4) return;
    const int c4 = C >> 2;
    const int col = (int)(i % c4);
    const int bi = (int)(i / ((long long)HW * c4));
    const int g = col / (c4 >> 5);
    const float mean = stat[2 * (bi * 32 + g)], rstd = stat[2 * (bi * 32 + g) + 1];
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[i];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[col], be = reinterpret_cast<const f32x4*>(beta)[col];
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = (xv[e] - mean) * rstd * ga[e] + be[e];
        if (swish) t = t / (1.0f + expf(-t));   // x * sigmoid(x)  (autoencoder.py:33-35)
        v[e] = t;
    }
    if constexpr (sizeof(T) == 4) {
        reinterpret_cast<f32x4*>(out)[i] = v;
    } else {
        T o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = Elem<T>::from_f32(v[e]);
        reinterpret_cast<uint2*>(out)[i] = *reinterpret_cast<const uint2*>(o4);
    }
}

// softmax over the last dim of S [rows, n] fp32 with a pre-scale; one wave per row; T out
template <typename T>
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ s, T* __restrict__ p, long long rows,
                                                           int n, float scale) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* sr = s + row * n;
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, sr[j] * scale);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += expf(sr[j] * scale - mx);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int j = lane; j < n; j += 64) p[row * n + j] = Elem<T>::from_f32(expf(sr[j] * scale - mx) * inv);
}

// out[i] = T(x[i])   (raw copy of the fp32 stream as a GEMM operand: the 1x1 shortcut convolution)
template <typename T>
__global__ void cast_kernel(const float* __restrict__ x, T* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Elem<T>::from_f32(x[i]);
}

inline dim3 grid1d(long long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

hipError_t launch_vae_input(const float* z, const float* w, const float* b, float inv_scale, float* out, int B, int HW,
                            hipStream_t s) {
    hipLaunchKernelGGL(vae_input_kernel, grid1d((long long)B * HW), dim3(256), 0, s, z, w, b, inv_scale, out, B, HW);
    return hipGetLastError();
}
hipError_t launch_vae_output(const float* in, float* out, int B, int C, int HW, int ldc, hipStream_t s) {
    hipLaunchKernelGGL(vae_output_kernel, grid1d((long long)B * C * HW), dim3(256), 0, s, in, out, B, C, HW, ldc);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_im2col3x3(const T* src, T* dst, int B, int H, int W, int C, int up, int Kpad, hipStream_t s) {
    constexpr int V = 16 / (int)sizeof(T);
    if (C % V || Kpad % V || Kpad < 9 * C) return hipErrorInvalidValue;
    const long long total = (long long)B * H * W * (Kpad / V);
    hipLaunchKernelGGL(im2col3x3_kernel<T>, grid1d(total), dim3(256), 0, s, src, dst, B, H, W, C, up, Kpad);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_im2col3x3_c4(const float* src, T* dst, int B, int H, int W, int Kpad, hipStream_t s) {
    hipLaunchKernelGGL(im2col3x3_c4_kernel<T>, grid1d((long long)B * H * W * Kpad), dim3(256), 0, s, src, dst, B, H, W, Kpad);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_groupnorm(const float* x, float* part, const float* gamma, const float* beta, T* out, int B, int HW, int C,
                            int swish, hipStream_t s) {
    if (C % 128 || C > 1024) return hipErrorInvalidValue;   // float4 columns: a multiple of 32, at most 256 per block
    const int chunks = (HW + kGnChunk - 1) / kGnChunk;
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * chunks), dim3(256), 0, s, x, part, HW, C, chunks);
    float* stat = part + (long long)B * chunks * 64;      // (mean, rstd) per (image, group) behind the partials
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * 32 + 255) / 256), dim3(256), 0, s, part, stat, B * 32, chunks,
                       (double)HW * (C / 32));
    const long long total4 = (long long)B * HW * (C / 4);
    hipLaunchKernelGGL(gn_apply_kernel<T>, grid1d(total4), dim3(256), 0, s, x, stat, gamma, beta, out, HW, C, swish, total4);
    return hipGetLastError();
}
int groupnorm_partials(int B, int HW) { return B * ((HW + kGnChunk - 1) / kGnChunk) * 32 * 2 + B * 64; }
template <typename T>
hipError_t launch_softmax_rows(const float* sc, T* p, long long rows, int n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(softmax_rows_kernel<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, sc, p, rows, n, scale);
    return hipGetLastError();
}
template <typename T>
hipError_t launch_cast(const float* x, T* out, long long n, hipStream_t s) {
    hipLaunchKernelGGL(cast_kernel<T>, grid1d(n), dim3(256), 0, s, x, out, n);
    return hipGetLastError();
}

#define DD_INST(T)                                                                                                     \
    template hipError_t launch_im2col3x3<T>(const T*, T*, int, int, int, int, int, int, hipStream_t);                 \
    template hipError_t launch_im2col3x3_c4<T>(const float*, T*, int, int, int, int, hipStream_t);                    \
    template hipError_t launch_groupnorm<T>(const float*, float*, const float*, const float*, T*, int, int, int, int, \
                                            hipStream_t);                                                             \
    template hipError_t launch_softmax_rows<T>(const float*, T*, long long, int, float, hipStream_t);                 \
    template hipError_t launch_cast<T>(const float*, T*, long long, hipStream_t);
DD_INST(bf16_t)
DD_INST(float)
#undef DD_INST

}  // namespace dd
