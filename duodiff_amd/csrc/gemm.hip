// MFMA GEMM for the U-ViT Linears on gfx950:  C[M,N] = A[M,K] . W[N,K]^T  (+ fused epilogue).
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights [out,in]), which is
// exactly the per-lane fragment order of v_mfma_f32_32x32x16_bf16 (8 consecutive k per lane)
// and, in the fp32 parity mode, of v_mfma_f32_32x32x2_f32 (one k per lane per instruction).
//
// Structure (one k-tile = 128 bytes of K per row: 64 bf16 or 32 fp32):
//   global --LDS-DMA (global_load_lds, 16 B/lane)--> LDS, two buffers, the load of k-tile
//   t+1 is in flight while the MFMAs of k-tile t run.  The LDS image is lane-linear (a
//   1 KiB wave-instruction = 8 rows x 128 B); bank conflicts of the ds_read_b128 fragment
//   reads are removed by XOR-swizzling the 16-byte chunk index with (row>>1)&7, applied to
//   the per-lane SOURCE address and again on the read (never to the LDS destination).
//   Accumulators stay in registers; the epilogue fuses bias / exact-erf GELU / residual add
//   into the store, so no Linear output makes an extra round trip through HBM.
//
// Replaces the ATen addmm/mm + gelu + add sequence of reference models/uvit.py:86-92,
// 155-168, 203-208.
#include "dd_internal.h"

namespace dd {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float v) {
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// Stage ROWS x 128 B into a lane-linear LDS tile with the source-side chunk swizzle.
// `g` points at (row 0, this k-tile's first byte); rows >= row_limit are clamped (N guard).
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const char* g, long long row_stride, char* lds_tile,
                                           int wave, int lane, int row_limit) {
    constexpr int PER_WAVE = ROWS / 8 / NWAVES;
#pragma unroll
    for (int j = 0; j < PER_WAVE; ++j) {
        const int inst = wave * PER_WAVE + j;
        int r = inst * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        r = r < row_limit ? r : row_limit - 1;
        glds16(g + (long long)r * row_stride + c * 16, lds_tile + inst * 1024);
    }
}

__device__ __forceinline__ f32x4 lds_read16(const char* p) {
    return *reinterpret_cast<const f32x4*>(p);
}

template <typename T>
__device__ __forceinline__ void mma_chunk(f32x16& acc, const f32x4& a, const f32x4& b);

template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(f32x16& acc, const f32x4& a, const f32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x16& acc, const f32x4& a, const f32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
}

// BM x BN output tile, WM x WN waves, each wave (BM/WM) x (BN/WN) as TM x TN MFMA 32x32 tiles.
template <typename T, int BM, int BN, int WM, int WN, int EPI>
__global__ void __launch_bounds__(WM* WN * 64)
gemm_kernel(const GemmArgs<T> a) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KT_ELEMS = 128 / (int)sizeof(T);  // k elements per k-tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // buffer b: A tile at b*STAGE_BYTES, then the W tile

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int h = lane >> 5, r32 = lane & 31;

    // tile mapping: consecutive blocks walk N first so that co-running blocks share A rows in L2
    const int n_tiles = (a.N + BN - 1) / BN;
    const int tile_m = blockIdx.x / n_tiles, tile_n = blockIdx.x % n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int n_limit = a.N - n0;  // rows of W valid in this tile (>= 1)

    const int nk = a.K / KT_ELEMS;
    const int nk1 = a.K1 / KT_ELEMS;
    const char* A1 = reinterpret_cast<const char*>(a.A) + (long long)m0 * a.lda * sizeof(T);
    const char* A2 = a.A2 ? reinterpret_cast<const char*>(a.A2) + (long long)m0 * a.lda2 * sizeof(T) : nullptr;
    const char* Wp = reinterpret_cast<const char*>(a.W) + (long long)n0 * a.K * sizeof(T);
    const long long sa1 = (long long)a.lda * sizeof(T), sa2 = (long long)a.lda2 * sizeof(T);
    const long long sw = (long long)a.K * sizeof(T);

    auto stage = [&](int kt, int buf) {
        char* at = smem + buf * STAGE_BYTES;
        if (kt < nk1) stage_tile<BM, NW>(A1 + (long long)kt * 128, sa1, at, wave, lane, BM);
        else stage_tile<BM, NW>(A2 + (long long)(kt - nk1) * 128, sa2, at, wave, lane, BM);
        stage_tile<BN, NW>(Wp + (long long)kt * 128, sw, at + A_BYTES, wave, lane, n_limit);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    stage(0, 0);
    __syncthreads();  // (emits vmcnt(0) for the LDS-DMA in flight)

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const char* Ab = smem + cur * STAGE_BYTES + (wr * (BM / WM)) * 128;
        const char* Bb = smem + cur * STAGE_BYTES + A_BYTES + (wc * (BN / WN)) * 128;
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            // logical 16-byte chunk this lane feeds: bf16 -> k = 16*step + 8h.. ; fp32 -> k = 16h + 4*step..
            const int chunk = sizeof(T) == 2 ? (2 * step + h) : (4 * h + step);
            f32x4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = i * 32 + r32;  // row within the wave's slab; swizzle uses the TILE row
                const int tr = wr * (BM / WM) + r;
                af[i] = lds_read16(Ab + r * 128 + ((chunk ^ ((tr >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = j * 32 + r32;
                const int tr = wc * (BN / WN) + r;
                bfr[j] = lds_read16(Bb + r * 128 + ((chunk ^ ((tr >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], af[i], bfr[j]);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * (BN / WN) + j * 32 + r32;
        const bool col_ok = col < a.N;
        float bias = 0.f;
        if (EPI != EPI_STORE && col_ok && a.bias) bias = a.bias[col];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (!col_ok || row >= a.M) continue;
                float v = acc[i][j][e];
                if (EPI == EPI_STORE) {
                    a.out[(long long)row * a.ldo + col] = Elem<T>::from_f32(v);
                } else if (EPI == EPI_BIAS_GELU) {
                    a.out[(long long)row * a.ldo + col] = Elem<T>::from_f32(gelu_erf(v + bias));
                } else if (EPI == EPI_BIAS_RESID) {
                    float* xp = a.xres + (long long)row * a.N + col;
                    v = *xp + (v + bias);
                    *xp = v;
                    if (a.out) a.out[(long long)row * a.ldo + col] = Elem<T>::from_f32(v);
                } else {  // EPI_BIAS_SET
                    a.xres[(long long)row * a.N + col] = v + bias;
                }
            }
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN>
hipError_t launch_cfg(const GemmArgs<T>& a, int epi, hipStream_t s) {
    const int m_tiles = (a.M + BM - 1) / BM, n_tiles = (a.N + BN - 1) / BN;
    const dim3 grid(m_tiles * n_tiles), block(WM * WN * 64);
    const size_t lds = 2 * (BM + BN) * 128;
#define DD_LAUNCH(E)                                                                              \
    {                                                                                             \
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, E>), grid, block, lds, s, a);          \
        return hipGetLastError();                                                                 \
    }
    switch (epi) {
        case EPI_STORE: DD_LAUNCH(EPI_STORE)
        case EPI_BIAS_GELU: DD_LAUNCH(EPI_BIAS_GELU)
        case EPI_BIAS_RESID: DD_LAUNCH(EPI_BIAS_RESID)
        case EPI_BIAS_SET: DD_LAUNCH(EPI_BIAS_SET)
    }
#undef DD_LAUNCH
    return hipErrorInvalidValue;
}

template <typename T, int BM, int BN, int WM, int WN>
hipError_t init_cfg() {
    const int lds = 2 * (BM + BN) * 128;
    hipError_t e = hipSuccess;
#define DD_ATTR(E)                                                                                         \
    if (e == hipSuccess)                                                                                   \
        e = hipFuncSetAttribute((const void*)gemm_kernel<T, BM, BN, WM, WN, E>,                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DD_ATTR(EPI_STORE) DD_ATTR(EPI_BIAS_GELU) DD_ATTR(EPI_BIAS_RESID) DD_ATTR(EPI_BIAS_SET)
#undef DD_ATTR
    return e;
}

}  // namespace

// dynamic-LDS opt-in for every instantiation, once per process (kept out of the launch path so
// that launches are capturable into a hipGraph)
hipError_t init_gemm_kernels() {
    hipError_t e = init_cfg<bf16_t, 128, 128, 2, 2>();
    if (e == hipSuccess) e = init_cfg<float, 128, 128, 2, 2>();
    return e;
}

template <typename T>
hipError_t launch_gemm(const GemmArgs<T>& a, int epilogue, hipStream_t s) {
    constexpr int KT = 128 / (int)sizeof(T);
    if (a.K % KT || a.K1 % KT || a.K1 > a.K || (a.K1 < a.K && !a.A2)) return hipErrorInvalidValue;
    return launch_cfg<T, 128, 128, 2, 2>(a, epilogue, s);
}

template hipError_t launch_gemm<bf16_t>(const GemmArgs<bf16_t>&, int, hipStream_t);
template hipError_t launch_gemm<float>(const GemmArgs<float>&, int, hipStream_t);

}  // namespace dd
