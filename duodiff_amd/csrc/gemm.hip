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

#include <type_traits>

namespace dd {

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

// exact-erf GELU (nn.GELU default), fp32 parity mode: libm erff.  (The bf16 mode uses the polynomial of gelu_erf4.)
__device__ __forceinline__ float gelu_erf_f32(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// The same piece in the scalar-base form: uniform 64-bit base (SGPRs) + a 32-bit lane offset, M0 = the piece's LDS address.  Written as asm: the builtin turns
// base + offset into a 64-bit VGPR address pair (checked in the ISA), and that form serialises with the SIMD's MFMAs (profiles/r05/dma_mfma_probe_roles.txt).
__device__ __forceinline__ void glds16s(const char* sbase, unsigned voff, const void* lds_dst) {
    const unsigned lds = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_dst;
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(voff), "s"(sbase) : "memory", "m0");
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

// GELU on a register quad; written on whole vectors so that the FMA chain maps onto packed fp32
// VALU ops (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of math per issue slot).
template <typename T>
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    if constexpr (sizeof(T) == 4) {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = gelu_erf_f32(v[e]);
        return r;
    } else {
        // bf16 mode: the result is rounded to bf16 (2^-9 relative), so erf(v/sqrt2) comes from an odd minimax polynomial
        // s*P(s^2) on the clamped argument s = med3(v, -3.8, 3.8) (|error| <= 1.3e-4; beyond the clamp erf is 1 - 1.4e-4):
        // gelu absolute error <= 2.4e-4, a quarter of the bf16 rounding of typical outputs.  No transcendental at all:
        // 4 v_med3 + 20 packed-fp32 ops per register quad.  The Abramowitz-Stegun 7.1.25 form used before (v_rcp + v_exp,
        // quarter rate) cost 2.4x more VALU time and made the fc1 epilogue ~29 us of pure VALU work per launch.
        f32x4 sc;
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[i] = __builtin_amdgcn_fmed3f(v[i], -3.8f, 3.8f);
        const f32x4 s2 = sc * sc;
        f32x4 p = s2 * 7.331517960e-08f + -4.544908101e-06f;
        p = p * s2 + 1.213693460e-04f;
        p = p * s2 + -1.863093246e-03f;
        p = p * s2 + 1.863326334e-02f;
        p = p * s2 + -1.314395642e-01f;
        p = p * s2 + 7.973534865e-01f;
        const f32x4 e = p * sc;
        const f32x4 hv = v * 0.5f;
        return hv * e + hv;
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

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// BM x BN output tile, WM x WN waves, each wave (BM/WM) x (BN/WN) as TM x TN MFMA 32x32 tiles.
// STAGES == 2: double buffer, one __syncthreads() per k-tile (drains the LDS-DMA each step).
// STAGES >= 3: LDS ring; the LDS-DMA of the next STAGES-2 k-tiles stays in flight ACROSS the
//   barrier: counted s_waitcnt vmcnt(N) + raw s_barrier, never __syncthreads() (it would emit
//   vmcnt(0)).  Order per k-tile: every wave waits for ITS pieces of tile t, barrier (now all
//   pieces of t have landed and everyone has finished reading tile t-1), re-stage the buffer of
//   tile t-1 with tile t+STAGES-1, then read tile t.
// XCD: blockIdx -> tile remap so that the blocks sharing an XCD (ids equal mod 8) get a contiguous
//   run of tiles, i.e. the column tiles that re-read one A row-slab hit the same L2 (speed only).
template <typename T, int BM, int BN, int WM, int WN, int STAGES, int EPI>
__global__ void __launch_bounds__(WM* WN * 64)
gemm_kernel(const GemmArgs<T> a) {
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int KT_ELEMS = 128 / (int)sizeof(T);  // k elements per k-tile
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;  // buffer b: A tile at b*STAGE_BYTES, then the W tile
    constexpr int G = (BM + BN) / 8 / NW;           // LDS-DMA wave-instructions per k-tile per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave / WN, wc = wave % WN;
    const int h = lane >> 5, r32 = lane & 31;

    // XCD-aware, bijective block -> tile map; tiles walk N fastest so neighbours share the A slab
    const int n_tiles = (a.N + BN - 1) / BN;
    int wg = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
    }
    const int tile_m = wg / n_tiles, tile_n = wg % n_tiles;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int n_limit = a.N - n0;  // rows of W valid in this tile (>= 1)
    const int m_limit = a.M - m0 < BM ? a.M - m0 : BM;   // rows of A valid in this tile (>= 1); the rest re-read the last one

    const int nk = a.K / KT_ELEMS;
    const int nk1 = a.K1 / KT_ELEMS;
    const char* A1 = reinterpret_cast<const char*>(a.A) + (long long)m0 * a.lda * sizeof(T);
    const char* A2 = a.A2 ? reinterpret_cast<const char*>(a.A2) + (long long)m0 * a.lda2 * sizeof(T) : nullptr;
    const char* Wp = reinterpret_cast<const char*>(a.W) + (long long)n0 * a.K * sizeof(T);
    const long long sa1 = (long long)a.lda * sizeof(T), sa2 = (long long)a.lda2 * sizeof(T);
    const long long sw = (long long)a.K * sizeof(T);

    auto stage = [&](int kt, int buf) {
        char* at = smem + buf * STAGE_BYTES;
        if (kt < nk1) stage_tile<BM, NW>(A1 + (long long)kt * 128, sa1, at, wave, lane, m_limit);
        else stage_tile<BM, NW>(A2 + (long long)(kt - nk1) * 128, sa2, at, wave, lane, m_limit);
        stage_tile<BN, NW>(Wp + (long long)kt * 128, sw, at + A_BYTES, wave, lane, n_limit);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
        const char* Ab = smem + buf * STAGE_BYTES + (wr * (BM / WM)) * 128;
        const char* Bb = smem + buf * STAGE_BYTES + A_BYTES + (wc * (BN / WN)) * 128;
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
                for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], bfr[j], af[i]);  // A-operand = W rows
        }
    };

    if constexpr (STAGES == 2) {
        stage(0, 0);
        __syncthreads();  // (emits vmcnt(0) for the LDS-DMA in flight)
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
            compute(cur);
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int s = 0; s < STAGES - 1; ++s)
            if (s < nk) stage(s, s);
        int rd = 0, wrb = STAGES - 1;  // ring slots: read slot of tile kt, write slot of tile kt+STAGES-1
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = nk - 1 - kt;  // tiles issued after kt that may stay in flight (capped)
            if (STAGES >= 4 && ahead >= 2) wait_vmcnt<2 * G>();
            else if (ahead >= 1) wait_vmcnt<G>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1, wrb);
            compute(rd);
            rd = rd + 1 == STAGES ? 0 : rd + 1;
            wrb = wrb + 1 == STAGES ? 0 : wrb + 1;
        }
    }

    // ---- epilogue.  The MFMA ran as D[n][m] = sum_k W[n][k] X[m][k] (weights as the A operand), so
    // in the 32x32 C/D map the LANE is the output row m (lane&31) and the 16 REGISTERS are output
    // columns n = (e&3) + 8*(e>>2) + 4*(lane>>5): each register quad is 4 consecutive columns of
    // one row -> one 8-byte (bf16) or 16-byte (fp32) access per quad instead of four scalar ones.
    // The T-typed output (bf16) additionally goes through a wave-private LDS strip, 32 rows at a
    // time, so that it leaves the CU as whole 16-byte-per-lane row segments (CW*2 contiguous bytes
    // per row) instead of 8-byte pieces scattered over 32 rows per instruction.
    constexpr int CW = BN / WN;                         // columns owned by one wave
    constexpr int STRIP_STRIDE = CW * (int)sizeof(T) + 16;  // padded row of the strip (16-B aligned)
    constexpr int LPR = CW * (int)sizeof(T) / 16;       // lanes per strip row on the way out
    constexpr int RPI = 64 / LPR;                       // rows per store instruction
    char* strip = smem + wave * (32 * STRIP_STRIDE);
    const bool use_out = EPI != EPI_BIAS_SET && a.out != nullptr;
    if (use_out) __syncthreads();                       // every wave is done with the operand tiles

    auto epilogue = [&](auto guard_tag) {
        constexpr bool GUARD = decltype(guard_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wr * (BM / WM) + i * 32 + r32;
            const bool row_ok = !GUARD || row < a.M;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int cw = j * 32 + 8 * g + 4 * h;      // column within the wave's span
                    const int col = n0 + wc * CW + cw;
                    const bool ok = !GUARD || (row_ok && col < a.N);
                    f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (ok) {
                        if (EPI != EPI_STORE && a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + col);
                        if (EPI == EPI_BIAS_GELU) v = gelu_erf4<T>(v);
                        if (EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_SET) {
                            f32x4* xp = reinterpret_cast<f32x4*>(a.xres + (long long)row * a.N + col);
                            if (EPI == EPI_BIAS_RESID) v = *xp + v;
                            *xp = v;
                        }
                    }
                    if (use_out) {
                        char* sp = strip + r32 * STRIP_STRIDE + cw * (int)sizeof(T);
                        if constexpr (sizeof(T) == 2) {
                            bf16_t o4[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o4[e] = f2bf(v[e]);
                            *reinterpret_cast<uint2*>(sp) = *reinterpret_cast<const uint2*>(o4);
                        } else {
                            *reinterpret_cast<f32x4*>(sp) = v;
                        }
                    }
                }
            }
            if (use_out) {
#pragma unroll
                for (int it = 0; it < 32 / RPI; ++it) {
                    const int sr = it * RPI + lane / LPR, sc = lane % LPR;   // strip row, 16-byte chunk
                    const f32x4 q = *reinterpret_cast<const f32x4*>(strip + sr * STRIP_STRIDE + sc * 16);
                    const int orow = m0 + wr * (BM / WM) + i * 32 + sr;
                    const int ocol = n0 + wc * CW + sc * (16 / (int)sizeof(T));
                    if (!GUARD || (orow < a.M && ocol < a.N)) {
                        const long long off = a.hm.L ? hm_offset(a.hm, orow, ocol) : (long long)orow * a.ldo + ocol;
                        *reinterpret_cast<f32x4*>(a.out + off) = q;
                    }
                }
            }
        }
    };
    // interior tiles (block-uniform test) skip every per-element bound check
    if (m0 + BM <= a.M && n0 + BN <= a.N) epilogue(std::false_type{});
    else epilogue(std::true_type{});
}

// ------------------------------------------------------------------------------------------
// Persistent 256x256 kernel (bf16): the production path for the big Linears.
//
// Why this shape: an ablation of the 128x128 kernel on MI355X showed global->LDS delivery tops
// out near 52 B/clk/CU, while a 128x128 tile needs 64 B/clk at full MFMA rate; 256x256 needs 32.
// Why persistent: K is only 512..2048 (8..32 k-tiles per tile), so per-tile prologue/epilogue is
// a large fraction; here the k-tile stream is continuous ACROSS tile boundaries -- while a tile's
// epilogue runs, the LDS-DMA of the next tile's first k-tile is already in flight and the
// epilogue's global stores stay in flight behind a counted vmcnt.
// Why the row partition: M = B*(256+extras) is never a multiple of 256 (257 tiles of 128 rows on
// 256 CUs is the worst quantisation there is).  Rows are independent, so each tile takes 256
// contiguous "main" rows plus `e` (<= 8) rows of the tail region [256*q, M): tile counts become
// q * N/256 -- for B = 128: 256 / 768 / 1024 tiles, exact multiples of the 256 CUs.  The tail rows
// ride along as one extra 32-row MFMA tile per wave (+12.5 % MFMA issue, no extra weight traffic).
//
// 8 waves = 2 (M) x 4 (N); wave tile 128 x 64 (4 x 2 MFMA 32x32x16 tiles) + 1 tail tile.
// LDS: 2 stages x (264 A rows + 256 W rows) x 128 B = 130 KB + two 1 KB bias slots.  The bf16 output leaves
// straight from registers (v_permlane32_swap pairs the two lane halves into 16-byte row segments).
// ------------------------------------------------------------------------------------------
struct Part256 {  // host-computed row partition
    int q;          // main M-tiles (256 rows each)
    int e;          // tail rows per tile (0..8)
    int tail_base;  // = 256 * q
};

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// four 16-byte LDS reads at addr + {0, 32, 64, 96} bytes, one wait
__device__ __forceinline__ void asm_ds_read_4x_b128_wait(unsigned addr, f32x4 (&r)[4]) {
    asm volatile(
        "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:32\n\tds_read_b128 %2, %4 offset:64\n\t"
        "ds_read_b128 %3, %4 offset:96\n\ts_waitcnt lgkmcnt(0)"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "v"(addr)
        : "memory");
}
constexpr int k256ARows = 264;                                   // 256 main + 8 tail rows
constexpr int k256Stage = (k256ARows + 256) * 128;               // 66560 B
constexpr int k256BiasOff = 2 * k256Stage;                       // two 1 KB bias slots (tile parity)
constexpr int k256Lds = k256BiasOff + 2 * 1024;                  // 135168 B

template <int EPI>
__global__ void __launch_bounds__(512)
gemm256_kernel(const GemmArgs<bf16_t> a, const Part256 part) {
    typedef bf16_t T;
    constexpr int TM = 4, TN = 2;
    constexpr int A_BYTES = k256ARows * 128;
    constexpr int N_EPI_STORES = (EPI == EPI_STORE || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_STORE) ? TM * 4 : TM * TN * 4;
    constexpr bool PARTIAL = EPI == EPI_PARTIAL;     // split-K: a work item = (tile, k range); the fp32 accumulators go to the split's slab as they are
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int h = lane >> 5, r32 = lane & 31;

    const int n_tiles = a.N >> 8;
    const int tiles_mn = part.q * n_tiles;
    const int S = PARTIAL ? a.splits : 1;
    const int total = tiles_mn * S;
    const int G = gridDim.x, gx = G >> 3, xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;  // G % 8 == 0
    auto tile_of = [&](int i) { return (i * 8 + xcd) * gx + wx; };
    int n_my = 0;
    while (tile_of(n_my) < total) ++n_my;
    if (n_my == 0) return;

    const int nk = a.K >> 6, nk1 = a.K1 >> 6;
    const long long sa1 = (long long)a.lda * 2, sw = (long long)a.K * 2;   // lda2 == lda (checked on the host)

    // LDS-DMA addressing: wave-uniform 64-bit base (SGPRs) + ONE 32-bit per-lane offset per operand.
    // Lane l of wave-instruction `inst` writes LDS row r = inst*8 + (l>>3), 16-byte slot l&7, and
    // must therefore fetch logical chunk (l&7) ^ ((r>>1)&7).  With r = 8*inst + lr that is
    // (l&7) ^ (lr>>1) for even inst and the same ^ 4 for odd inst, i.e. byte offset ^ 64: per-lane
    // address state is two VGPRs for the whole kernel (it used to be ~18 and got spilled).
    auto stage = [&](int tm, int tn, int kt, int buf, int parity) {
        // recomputed per call from an opaque copy of the lane id (4 VALU ops): kept live across the
        // k-loop these offsets were the registers the allocator chose to spill
        unsigned l = (unsigned)lane;
        asm volatile("" : "+v"(l));
        const unsigned lr = l >> 3;
        const unsigned swz = ((l & 7u) ^ (lr >> 1)) << 4;
        const unsigned voff_a = lr * (unsigned)sa1 + swz;
        const unsigned voff_w = lr * (unsigned)sw + swz;
        char* at = smem + buf * k256Stage;
        if (kt == 0 && wave == 1 && EPI != EPI_STORE && !PARTIAL && a.bias)   // this tile's 256 bias values -> LDS
            glds16(a.bias + tn * 256 + lane * 4, at - buf * k256Stage + k256BiasOff + parity * 1024);
        const char* Ab = kt < nk1 ? reinterpret_cast<const char*>(a.A) + (long long)kt * 128
                                  : reinterpret_cast<const char*>(a.A2) + (long long)(kt - nk1) * 128;
        const char* a_rows = Ab + ((long long)tm * 256 + wave * 32) * sa1;     // sa1 == sa2 (checked on host)
        const char* w_rows = reinterpret_cast<const char*>(a.W) + ((long long)tn * 256 + wave * 32) * sw + (long long)kt * 128;
#pragma unroll
        for (int inst = 0; inst < 4; ++inst) {
            const unsigned x = (inst & 1) ? 64u : 0u;
            glds16s(a_rows + (long long)(inst * 8) * sa1, voff_a ^ x, at + (wave * 4 + inst) * 1024);
        }
        if (wave == 0 && part.e > 0) {
            // 8 tail rows of this tile -> LDS rows 256..263 (row index 256 + lr: even-inst swizzle)
            long long row = (long long)part.tail_base + (long long)tm * part.e + ((int)lr < part.e ? (int)lr : part.e - 1);
            row = row < a.M ? row : a.M - 1;
            glds16(Ab + row * sa1 + swz, at + 256 * 128);
        }
#pragma unroll
        for (int inst = 0; inst < 4; ++inst) {
            const unsigned x = (inst & 1) ? 64u : 0u;
            glds16s(w_rows + (long long)(inst * 8) * sw, voff_w ^ x, at + A_BYTES + (wave * 4 + inst) * 1024);
        }
    };

    f32x16 acc[TM][TN], accx;
    // Fragment loads are software-pipelined: the 7 ds_read_b128 of k-step s+1 are issued between the
    // 9 MFMAs of k-step s (sched_group_barrier pins that interleave), so only the first step of a
    // k-tile exposes LDS latency.
    // LDS fragment addresses: for lane (row tr, half h) the k-step s reads chunk (2s+h) ^ f(tr), and
    // (2s+h) ^ f == ((h ^ f)) ^ 2s, so address(s) = address(0) ^ (s << 5): ONE offset register per
    // fragment (7 in all) plus an XOR per read, instead of 28 hoisted address registers.
    struct Frags { f32x4 a[TM], b[TN], x; };
    unsigned a_off[TM], b_off[TN], x_off;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int tr = wr * 128 + i * 32 + r32;
        a_off[i] = tr * 128 + ((h ^ ((tr >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int tr = wc * 64 + j * 32 + r32;
        b_off[j] = A_BYTES + tr * 128 + ((h ^ ((tr >> 1) & 7)) << 4);
    }
    {
        const int tr = 256 + r32;  // tail rows; rows >= 264 read the W tile's bytes: finite or not, they
        x_off = tr * 128 + ((h ^ ((tr >> 1) & 7)) << 4);  // only reach dropped output columns
    }
    const unsigned smem_lds = lds_addr(smem);
    typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
    // Called once per k-tile: makes the 7 offsets opaque to the optimiser so that it cannot hoist the
    // 28 (offset ^ step) values out of the k-loop again (they would be spilled to scratch).
    auto pin_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a_off[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(b_off[j]));
        asm volatile("" : "+v"(x_off));
    };
    auto load_frags = [&](unsigned stage_lds, int step, Frags& f) {
        const unsigned sx = (unsigned)step << 5;
#pragma unroll
        for (int j = 0; j < TN; ++j) f.b[j] = *(lds_f32x4_ptr)(size_t)(stage_lds + (b_off[j] ^ sx));
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a[i] = *(lds_f32x4_ptr)(size_t)(stage_lds + (a_off[i] ^ sx));
        f.x = *(lds_f32x4_ptr)(size_t)(stage_lds + (x_off ^ sx));
    };
    auto mma_step = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma_chunk<T>(acc[i][j], f.b[j], f.a[i]);  // A-operand = W rows
        mma_chunk<T>(accx, wr == 0 ? f.b[0] : f.b[1], f.x);
    };
    // 9 MFMAs of this k-step with the 7 fragment reads of the next one packed EARLY (2 reads behind each of the
    // first MFMAs): the lgkmcnt(0) in front of the next step is then covered by 5 MFMAs instead of 2.
    auto interleave = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 DS reads
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
    };
    // `prefetch` (the LDS-DMA of the next k-tile) is issued AFTER this k-tile's first fragment reads, so their
    // latency overlaps the 9-17 DMA instructions instead of following them.
    auto compute = [&](int buf, auto&& prefetch) {
        const unsigned st = smem_lds + buf * k256Stage;
        pin_offsets();
        Frags f0, f1;
        load_frags(st, 0, f0);
        __builtin_amdgcn_sched_barrier(0);   // step 0's own fragments stay ahead of the DMA issue and the MFMA region
        prefetch();
        __builtin_amdgcn_sched_barrier(0);
        load_frags(st, 1, f1);
        mma_step(f0);
        interleave();
        load_frags(st, 2, f0);
        mma_step(f1);
        interleave();
        load_frags(st, 3, f1);
        mma_step(f0);
        interleave();
        mma_step(f1);
    };

    // Epilogue.  Two rules keep it from serialising on memory latency:
    //  * no global LOAD may sit behind a global STORE of the same wave (vmcnt retires in order, so
    //    a load's wait would also wait for every older store): the bias comes from LDS, and the
    //    residual x is software-pipelined -- the loads of unit u+1 are issued BEFORE the stores of
    //    unit u, so waiting for them only retires stores that are two units old;
    //  * the bf16 copy leaves as 16-byte row segments built in registers (v_permlane32_swap), no LDS round trip.
    auto pack4 = [](const f32x4& q) -> uint2 {   // two v_cvt_pk_bf16_f32, no mask/shift glue
        typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
        const bf16v4 b = __builtin_convertvector(q, bf16v4);
        return __builtin_bit_cast(uint2, b);
    };
    constexpr bool HAS_BIAS = EPI != EPI_STORE && !PARTIAL;
    constexpr bool RESID = EPI == EPI_BIAS_RESID;
    constexpr bool WRITES_X = EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_SET;

    auto epilogue = [&](int lin, int parity, int sp) {
        const int tm = lin / n_tiles, tn = lin - tm * n_tiles;
        float* const xbase = PARTIAL ? a.partial + (long long)sp * a.M * a.N : a.xres;     // (split-K: this split's [M, N] slab)
        const bool use_out = EPI != EPI_BIAS_SET && !PARTIAL && a.out != nullptr;
        const bool has_bias = HAS_BIAS && a.bias != nullptr;
        const int col0 = tn * 256 + wc * 64;
        const unsigned bias_lds = lds_addr(smem + k256BiasOff + parity * 1024) + (wc * 64 + 4 * h) * 4;
        // bias quads of column tile j: columns j*32 + 8g + 4h.., i.e. LDS bytes +32 per g
        auto load_bias = [&](int j, f32x4 (&b)[4]) {
            if (has_bias) asm_ds_read_4x_b128_wait(bias_lds + j * 128, b);
            else b[0] = b[1] = b[2] = b[3] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        const long long row_base = (long long)tm * 256 + wr * 128 + r32;
        auto xptr = [&](int i, int j, int g) {
            return reinterpret_cast<f32x4*>(xbase + (row_base + i * 32) * a.N + col0 + j * 32 + 8 * g + 4 * h);
        };
        f32x4 xl[2][4];   // residual quads of unit u = 2*i + j, double-buffered
        if (RESID) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xl[0][g] = *xptr(0, 0, g);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            uint2 v[TN][4];   // bf16-packed results of this 32-row slab
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int u = i * TN + j;
                f32x4 bias[4];
                if (HAS_BIAS) load_bias(j, bias);
                if (RESID && u + 1 < TM * TN) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) xl[(u + 1) & 1][g] = *xptr((u + 1) / TN, (u + 1) % TN, g);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 q = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                    if (HAS_BIAS) q += bias[g];
                    if (EPI == EPI_BIAS_GELU) q = gelu_erf4<T>(q);
                    if (RESID) q = xl[u & 1][g] + q;
                    if (WRITES_X || PARTIAL) *xptr(i, j, g) = q;
                    if (!PARTIAL) v[j][g] = pack4(q);
                }
            }
            if (use_out) {
                // bf16 copy of this 32-row slab straight from registers: quads g and g+1 of a column block hold, on lane (r, h=0),
                // columns 8g..8g+3 and 8g+8..8g+11 of row r and on lane (r, h=1) columns 8g+4..8g+7 and 8g+12..8g+15;
                // v_permlane32_swap (upper half of the first register <-> lower half of the second) makes that 16 contiguous
                // bytes per lane -- columns 8g..8g+7 on h=0, 8g+8..8g+15 on h=1 -- so the slab leaves as 16-byte stores without
                // the LDS strip round trip (and its lgkmcnt(0) waits) the first version of this epilogue needed.
                const long long orow = (long long)tm * 256 + wr * 128 + i * 32 + r32;
                // head-major (qkv): the wave's 64 columns are one (q | k | v, head) unit -- the 32 rows of this slab are
                // 32 consecutive 128-byte rows of that unit (4 KB contiguous, unless the slab crosses an image boundary)
                T* obase = a.hm.L ? a.out + hm_offset(a.hm, (int)orow, col0) : a.out + orow * a.ldo + col0;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int gp = 0; gp < 4; gp += 2) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(v[j][gp].x, v[j][gp + 1].x, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(v[j][gp].y, v[j][gp + 1].y, false, false);
                        const uint4 o = {s0[0], s1[0], s0[1], s1[1]};
                        *reinterpret_cast<uint4*>(obase + j * 32 + 8 * gp + 8 * h) = o;
                    }
            }
        }
        // tail rows: lane = tail row index (valid below part.e), registers = this wave's 32 columns
        if (part.e > 0) {
            const long long row = (long long)part.tail_base + (long long)tm * part.e + r32;
            f32x4 bias[4];
            if (HAS_BIAS) load_bias(wr, bias);
            if (r32 < part.e && row < a.M) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = col0 + wr * 32 + 8 * g + 4 * h;
                    f32x4 q = {accx[4 * g], accx[4 * g + 1], accx[4 * g + 2], accx[4 * g + 3]};
                    if (HAS_BIAS) q += bias[g];
                    if (EPI == EPI_BIAS_GELU) q = gelu_erf4<T>(q);
                    if (WRITES_X || PARTIAL) {
                        f32x4* xp = reinterpret_cast<f32x4*>(xbase + row * a.N + col);
                        if (RESID) q = *xp + q;
                        *xp = q;
                    }
                    if (use_out) {
                        const long long off = a.hm.L ? hm_offset(a.hm, (int)row, col) : row * a.ldo + col;
                        *reinterpret_cast<uint2*>(a.out + off) = pack4(q);
                    }
                }
            }
        }
    };

    // work item i of this workgroup: tile (tm, tn) = lin, k-tiles [kb, ke) (the whole k range unless split-K)
    struct Item { int lin, tm, tn, sp, kb, ke; };
    auto item_of = [&](int i) {
        const int it = tile_of(i);
        Item w;
        w.sp = it / tiles_mn;
        w.lin = it - w.sp * tiles_mn;
        w.tm = w.lin / n_tiles;
        w.tn = w.lin - w.tm * n_tiles;
        w.kb = w.sp * nk / S;
        w.ke = (w.sp + 1) * nk / S;
        return w;
    };
    Item cur = item_of(0), nxt = cur;
    stage(cur.tm, cur.tn, cur.kb, 0, 0);
    int buf = 0;
    bool stores_in_flight = false;
    for (int i = 0; i < n_my; ++i) {
        if (i + 1 < n_my) nxt = item_of(i + 1);
#pragma unroll
        for (int ti = 0; ti < TM; ++ti)
#pragma unroll
            for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ti][tj][e] = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) accx[e] = 0.f;
        for (int kt = cur.kb; kt < cur.ke; ++kt) {
            // the k-tile about to be read must have landed; only an epilogue's stores may be younger
            if (kt == cur.kb && stores_in_flight) wait_vmcnt<N_EPI_STORES>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            compute(buf, [&]() {
                if (kt + 1 < cur.ke) stage(cur.tm, cur.tn, kt + 1, buf ^ 1, i & 1);
                else if (i + 1 < n_my) stage(nxt.tm, nxt.tn, nxt.kb, buf ^ 1, (i + 1) & 1);
            });
            buf ^= 1;
        }
        epilogue(cur.lin, i & 1, cur.sp);
        stores_in_flight = true;
        cur = nxt;
    }
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES>
hipError_t launch_cfg(const GemmArgs<T>& a, int epi, hipStream_t s) {
    const int m_tiles = (a.M + BM - 1) / BM, n_tiles = (a.N + BN - 1) / BN;
    const dim3 grid(m_tiles * n_tiles), block(WM * WN * 64);
    const size_t lds = (size_t)STAGES * (BM + BN) * 128;
#define DD_LAUNCH(E)                                                                              \
    {                                                                                             \
        hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, STAGES, E>), grid, block, lds, s, a);  \
        return hipGetLastError();                                                                 \
    }
    switch (epi) {
        case EPI_STORE: DD_LAUNCH(EPI_STORE)
        case EPI_BIAS_GELU: DD_LAUNCH(EPI_BIAS_GELU)
        case EPI_BIAS_RESID: DD_LAUNCH(EPI_BIAS_RESID)
        case EPI_BIAS_SET: DD_LAUNCH(EPI_BIAS_SET)
        case EPI_BIAS_STORE: DD_LAUNCH(EPI_BIAS_STORE)
    }
#undef DD_LAUNCH
    return hipErrorInvalidValue;
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES>
hipError_t init_cfg() {
    const int lds = STAGES * (BM + BN) * 128;
    hipError_t e = hipSuccess;
#define DD_ATTR(E)                                                                                         \
    if (e == hipSuccess)                                                                                   \
        e = hipFuncSetAttribute((const void*)gemm_kernel<T, BM, BN, WM, WN, STAGES, E>,                    \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    DD_ATTR(EPI_STORE) DD_ATTR(EPI_BIAS_GELU) DD_ATTR(EPI_BIAS_RESID) DD_ATTR(EPI_BIAS_SET) DD_ATTR(EPI_BIAS_STORE)
#undef DD_ATTR
    return e;
}

// Row partition for gemm256 (see the kernel's header): q main tiles, e tail rows per tile, chosen to
// minimise rounds over the CUs; returns false when the shape does not fit the kernel.
bool plan256(int M, int N, int K, int K1, int num_cus, Part256& p) {
    if (N % 256 || K % 64 || K1 % 64 || M < 256 || (long long)K * 2 * 8 >= (1ll << 31)) return false;
    const int nt = N / 256;
    const int q_hi = M / 256, q_lo = (M + 263) / 264;
    double best = 1e30;
    int best_q = -1;
    for (int q = q_hi; q >= q_lo && q >= 1; --q) {
        const int tail = M - 256 * q;
        const int e = tail > 0 ? (tail + q - 1) / q : 0;
        if (e > 8) continue;
        const long long tiles = (long long)q * nt;
        const double rounds = (double)((tiles + num_cus - 1) / num_cus);
        const double cost = rounds * (e > 0 ? 1.125 : 1.0);
        if (cost < best - 1e-9) { best = cost; best_q = q; }
    }
    if (best_q < 0) return false;
    p.q = best_q;
    const int tail = M - 256 * best_q;
    p.e = tail > 0 ? (tail + best_q - 1) / best_q : 0;
    p.tail_base = 256 * best_q;
    return true;
}

hipError_t launch_256(const GemmArgs<bf16_t>& a, int epi, const Part256& p, int num_cus, hipStream_t s) {
    const int tiles = p.q * (a.N / 256) * (epi == EPI_PARTIAL ? a.splits : 1);
    int grid = num_cus;                             // a multiple of 8: the kernel groups workgroups by XCD (gx = G >> 3)
    if (tiles < grid) grid = (tiles + 7) / 8 * 8;
#define DD_LAUNCH(E)                                                                                  \
    {                                                                                                 \
        hipLaunchKernelGGL((gemm256_kernel<E>), dim3(grid), dim3(512), k256Lds, s, a, p);             \
        return hipGetLastError();                                                                     \
    }
    switch (epi) {
        case EPI_STORE: DD_LAUNCH(EPI_STORE)
        case EPI_BIAS_GELU: DD_LAUNCH(EPI_BIAS_GELU)
        case EPI_BIAS_RESID: DD_LAUNCH(EPI_BIAS_RESID)
        case EPI_BIAS_SET: DD_LAUNCH(EPI_BIAS_SET)
        case EPI_BIAS_STORE: DD_LAUNCH(EPI_BIAS_STORE)
        case EPI_PARTIAL: DD_LAUNCH(EPI_PARTIAL)
    }
#undef DD_LAUNCH
    return hipErrorInvalidValue;
}

}  // namespace

// dynamic-LDS opt-in for every instantiation, once per process (kept out of the launch path so
// that launches are capturable into a hipGraph)
hipError_t init_gemm_kernels() {
    hipError_t e = init_cfg<float, 128, 128, 2, 2, 2>();
    if (e == hipSuccess) e = init_cfg<float, 128, 64, 2, 2, 2>();
    if (e == hipSuccess) e = init_cfg<bf16_t, 128, 128, 2, 2, 2>();
#define DD_ATTR(E)                                                                                  \
    if (e == hipSuccess)                                                                            \
        e = hipFuncSetAttribute((const void*)gemm256_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, k256Lds);
    DD_ATTR(EPI_STORE) DD_ATTR(EPI_BIAS_GELU) DD_ATTR(EPI_BIAS_RESID) DD_ATTR(EPI_BIAS_SET) DD_ATTR(EPI_BIAS_STORE) DD_ATTR(EPI_PARTIAL)
#undef DD_ATTR
    return e;
}

// CU count the persistent grids are sized for: the device's, rounded down to a multiple of 8 (the kernel assumes it)
int device_num_cus() {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
        n = prop.multiProcessorCount;
    return n / 8 * 8;
}

bool plan_rows_256(int M, int N, int K, int num_cus, int* q, int* e) {
    Part256 p{};
    const bool ok = plan256(M, N, K, K, num_cus >= 8 ? num_cus / 8 * 8 : 256, p);
    if (ok) { *q = p.q; *e = p.e; }
    return ok;
}

// Split-K form of the 256 x 256 kernel for Linears with few output tiles (embed_dim-wide outputs at small batches: 32 x 4 tiles at
// ImageNet-256 latents, B = 32): a work item = (tile, one of a.splits equal k ranges), its fp32 accumulators are stored as they are to
// slab a.partial[split][M][N]; launch_reduce_ln adds the slabs in ascending split order (+ bias, residual, LayerNorm).  The split is a
// function of the Linear's shape alone, never of the batch.
bool gemm_splitk_supported(int M, int N, int K, int K1, int splits) {
    Part256 p;
    return splits >= 2 && (K / 64) % splits == 0 && plan256(M, N, K, K1, 256, p);
}
hipError_t launch_gemm_splitk(const GemmArgs<bf16_t>& a, hipStream_t s, int num_cus) {
    Part256 p;
    const int cus = num_cus >= 8 ? num_cus / 8 * 8 : 256;
    if (!a.partial || a.hm.L || !gemm_splitk_supported(a.M, a.N, a.K, a.K1, a.splits) || (a.K1 != a.K && (a.lda != a.lda2 || !a.A2)) ||
        !plan256(a.M, a.N, a.K, a.K1, cus, p))
        return hipErrorInvalidValue;
    return launch_256(a, EPI_PARTIAL, p, cus, s);
}

// a launch whose 256 x 256 tiles would leave half of the grid's CUs without one (ImageNet-256 latents, B = 32: the N = 1024 Linears have
// 32 x 4 tiles) is better off with the 128 x 128 kernel: four times the tiles, two workgroups per CU
bool gemm_prefers_128(int M, int N, int K, int K1, int num_cus) {
    Part256 p;
    const int cus = num_cus >= 8 ? num_cus / 8 * 8 : 256;
    return plan256(M, N, K, K1, cus, p) && (long long)p.q * (N / 256) * 2 <= cus;
}

// bf16: the persistent 256x256 kernel where the shape fits it, else the generic 128x128 kernel; fp32 (parity mode, exact
// f32 MFMA): generic kernel.  num_cus sizes the persistent grid (per context: dd_set_num_cus).
template <typename T>
hipError_t launch_gemm(const GemmArgs<T>& a, int epilogue, hipStream_t s, int num_cus) {
    constexpr int KT = 128 / (int)sizeof(T);
    if (a.K % KT || a.K1 % KT || a.K1 > a.K || (a.K1 < a.K && !a.A2)) return hipErrorInvalidValue;
    if (a.hm.L && (a.N % 64 || a.N != 3 * 64 * a.hm.H || (long long)a.M * a.hm.L >= (1ll << 32))) return hipErrorInvalidValue;
    if constexpr (sizeof(T) == 4) {
        if (a.N <= 64) return launch_cfg<T, 128, 64, 2, 2, 2>(a, epilogue, s);  // decoder_pred: N = P*P*C <= 64
        return launch_cfg<T, 128, 128, 2, 2, 2>(a, epilogue, s);
    } else {
        Part256 p;
        const int cus = num_cus >= 8 ? num_cus / 8 * 8 : 256;
        if ((a.K1 == a.K || a.lda == a.lda2) && plan256(a.M, a.N, a.K, a.K1, cus, p)) {
            const bool small = a.tile128 < 0 ? (long long)p.q * (a.N / 256) * 2 <= cus : a.tile128 == 1;     // (gemm_prefers_128; the caller's decision if it made one)
            if (small && !a.hm.L) return launch_cfg<T, 128, 128, 2, 2, 2>(a, epilogue, s);
            return launch_256(a, epilogue, p, cus, s);
        }
        return launch_cfg<T, 128, 128, 2, 2, 2>(a, epilogue, s);   // shapes the 256x256 kernel does not take (small N, tiny M)
    }
}

template hipError_t launch_gemm<bf16_t>(const GemmArgs<bf16_t>&, int, hipStream_t, int);
template hipError_t launch_gemm<float>(const GemmArgs<float>&, int, hipStream_t, int);

}  // namespace dd
