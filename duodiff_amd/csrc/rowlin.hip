// Row-resident Linear + bias + residual + LayerNorm for embed_dim 768 (ImageNet-64), bf16 MFMA operands, fp32 accumulation:
//     x  += A . W^T + b            (attn.proj: A = attention output, K = 768;  fc2: A = GELU(fc1(.)), K = 3072; reference models/uvit.py:166, 91, 206-207)
//     h   = LayerNorm(x) * g + beta   as bf16 (the next Linear's operand: norm2 after attn.proj, the next block's norm1 after fc2)
// The GEMM sequence runs these as a 256 x 256-tile GEMM whose epilogue read-modify-writes the fp32 residual rows 32 bytes at a time, followed
// by a LayerNorm launch that reads the rows again.  Here a wave keeps its 32 rows x 768 outputs in accumulators (384 registers: one wave
// per SIMD) for the whole k range, so the residual rows are read once and written once (whole 16-byte quads of a lane's row), the
// LayerNorm comes from the registers, and no LayerNorm launch exists.  At embed_dim 1024 the outputs alone are the whole register file.
//   * workgroup = 128 rows (4 waves x 32); k-outer: per k-step (16 k) the 24 output tiles' weight fragments (24 KB, host-packed in stream
//     order) come through a ring of four LDS slots by LDS-DMA, requested three k-steps ahead; the rows' operand A goes through LDS too
//     (16 KB per 4 k-steps: 128 rows x 128 B in gemm256's XOR-swizzled row layout, three buffers), so every vector-memory request of
//     the loop is an LDS-DMA piece in a fixed order and the waits are exact counts;
//   * one barrier per k-step (24 MFMAs per wave);
//   * rows as in the fused block tail: the PATCH rows of all images form the main row space, cut into tiles of 128 (a wave's 32 rows = one
//     fragment group of the attention launch, so the next block's norm1 leaves in fragment order); the few EXTRA-token rows are gathered into
//     tiles of their own, each split along K over `groups` workgroups that write partial sums to slabs -- mlp_reduce_kernel folds the slabs
//     into x in a fixed order (deterministic; which path a row takes depends only on its token index, never on the batch size).
#include "dd_internal.h"

#include <utility>

namespace dd {
namespace {

typedef const __attribute__((address_space(1))) void* rl_gptr_t;
typedef __attribute__((address_space(3))) void* rl_lptr_t;

constexpr int kRlD = 768, kRlNT = kRlD / 32, kRlBlk = kRlNT * 1024;      // one k-step of weights: 24 fragments
constexpr int kRlRing = 4 * kRlBlk;                                       // 96 KB
constexpr int kRlABuf = 128 * 128;                                        // 128 rows x 64 k (4 k-steps) bf16
constexpr int kRlLds = kRlRing + 3 * kRlABuf;                             // 144 KB

__device__ __forceinline__ unsigned rl_pack2(float lo, float hi) {
    typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
    typedef float f32v2 __attribute__((ext_vector_type(2)));
    const f32v2 q = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16v2));
}

__device__ __forceinline__ unsigned rl_lds(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
// one 1 KB LDS-DMA piece in the scalar-base form: uniform 64-bit base (SGPRs) + a 32-bit lane offset, M0 = the piece's LDS address.  Written as asm: the
// builtin turns base + offset into a 64-bit VGPR address pair, and that form serialises with the SIMD's MFMAs (profiles/r05/dma_mfma_probe_roles.txt).
__device__ __forceinline__ void rl_dma(const char* sbase, unsigned voff, const void* lds_dst) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(rl_lds(lds_dst)), "v"(voff), "s"(sbase) : "memory", "m0");
}
template <int OFF>
__device__ __forceinline__ void rl_read(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
// one MFMA gap: [wait: at most LG LDS reads younger than this MFMA's fragment] MFMA [read the fragment PD gaps ahead into the register just read]
// (AGPR: the accumulator lives in the AGPR half of the register file -- 256 registers = 16 of the 24 output tiles; the other 8 tiles' in VGPRs)
template <int LG, int LO, bool AGPR>
__device__ __forceinline__ void rl_gap(f32x16& acc, bf16x8& wa, const bf16x8& xb, unsigned la) {
    if constexpr (AGPR)
        asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%5"
                     : "+a"(acc), "+v"(wa) : "v"(xb), "v"(la), "i"(LG), "i"(LO));
    else
        asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%5"
                     : "+v"(acc), "+v"(wa) : "v"(xb), "v"(la), "i"(LG), "i"(LO));
}
constexpr int kRlNA = 16;        // output tiles whose accumulators are AGPRs
template <int T>
__device__ __forceinline__ void rl_pin(f32x16& y) {
    if constexpr (T < kRlNA) asm volatile("" : "+a"(y));
    else asm volatile("" : "+v"(y));
}

__global__ void __launch_bounds__(256) rowlin768_kernel(const RowLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* abuf = smem + kRlRing;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    // tile kinds: plain (a.tok_n == 0: rows [0, M) in tiles of 128), main (patch rows), extras (extra-token rows, a k-slice of the Linear)
    const bool planned = a.tok_n > 0;
    const bool part_tile = planned && (int)blockIdx.x >= a.tiles_main;
    const int xe = part_tile ? (int)blockIdx.x - a.tiles_main : 0;
    const int tile = part_tile ? xe / a.groups : (int)blockIdx.x;
    const int ks0 = part_tile ? (xe % a.groups) * a.cpg : 0;                     // first k-step of this workgroup's range (a multiple of 4)
    const int ks_all = a.K >> 4;
    const int KS = part_tile ? (ks0 + a.cpg < ks_all ? a.cpg : ks_all - ks0) : ks_all, NG = KS >> 2;   // k-steps of the range, groups of 4
    // logical row idx of the tile -> token row (clamped to a valid row; stores are masked)
    auto row_of = [&](int local, bool& ok) -> long long {
        const int idx = tile * 128 + local;
        if (!planned) { ok = idx < a.M; return ok ? idx : a.M - 1; }
        if (!part_tile) {
            ok = idx < a.n_main;
            const int p = ok ? idx : a.n_main - 1, b = p / a.tok_n;
            return (long long)b * a.tok_l + a.tok_e + (p - b * a.tok_n);
        }
        ok = idx < a.n_extra;
        const int q = ok ? idx : a.n_extra - 1, b = q / a.tok_e;
        return (long long)b * a.tok_l + (q - b * a.tok_e);
    };

    // LDS-DMA of the weights of k-step ks -> ring slot ks & 3: 24 pieces, 6 per wave
    const char* wsrc = a.wimg + (size_t)ks0 * kRlBlk;
    const unsigned lane16 = lane * 16;
    auto dma_w = [&](int ks) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int pc = wave * 6 + i;
            rl_dma(wsrc + (size_t)ks * kRlBlk + pc * 1024, lane16, ring + (ks & 3) * kRlBlk + pc * 1024);
        }
    };
    // LDS-DMA of the rows' operand, group g (k = 64 g .. 64 g + 63) -> A buffer g % 3: 16 pieces of 8 rows x 128 B, 4 per wave; the lane
    // fetches the 16-byte chunk that belongs in its slot (source-side XOR swizzle: slot s of row r holds chunk s ^ ((r >> 1) & 7))
    const int lr = lane >> 3;
    unsigned arows[4];          // this lane's four operand rows (one per piece) at the first k of the range: byte offsets (launch_rowlin checks 4 GB)
    // cat[A | A2] along k (skip_linear): groups of 64 k at or behind k_split read A2 (shifted so that the lane offsets, which carry the absolute k, still apply)
    const char* abase_1 = reinterpret_cast<const char*>(a.A);
    const char* abase_2 = a.k_split > 0 ? reinterpret_cast<const char*>(a.A2) - (long long)a.k_split * 2 : abase_1;
    const int g_split = a.k_split > 0 ? (a.k_split >> 6) - (ks0 >> 2) : (1 << 30);       // first group (relative to this workgroup's range) that reads A2
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bool okr;
        const int r = (wave * 4 + i) * 8 + lr;
        arows[i] = (unsigned)((row_of(r, okr) * a.lda + (long long)ks0 * 16) * 2) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
    }
    auto dma_a = [&](int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = wave * 4 + i;
            rl_dma(g < g_split ? abase_1 : abase_2, arows[i] + (unsigned)g * 128u, abuf + (g % 3) * kRlABuf + pc * 1024);
        }
    };
    // ... and one piece at a time (the loop spreads a k-step's requests over its MFMA gaps: an LDS-DMA instruction costs 60-185 cycles of
    // issue, a burst of 10 stalls the wave for a whole k-step's worth of MFMA time)
    auto dma_w1 = [&](int ks, int j) {
        const int pc = wave * 6 + j;
        rl_dma(wsrc + (size_t)ks * kRlBlk + pc * 1024, lane16, ring + (ks & 3) * kRlBlk + pc * 1024);
    };
    auto dma_a1 = [&](int g, int j) {
        const int pc = wave * 4 + j;
        rl_dma(g < g_split ? abase_1 : abase_2, arows[j] + (unsigned)g * 128u, abuf + (g % 3) * kRlABuf + pc * 1024);
    };
    // prologue: A groups 0, 1; weights of k-steps 0, 1, 2
    dma_a(0);
    dma_a(1);
    dma_w(0);
    dma_w(1);
    dma_w(2);

    // Accumulators: 384 registers resident for the whole kernel (tiles 0-15 in the AGPR half of the register file, 16-23 in VGPRs); the loop is one asm volatile
    // statement per MFMA gap (MFMA + the LDS read of the weight fragment PD gaps ahead, into the register the MFMA just read + a counted
    // lgkmcnt), as the fused block tail's projection phases: hipcc neither keeps 384 accumulators in place through builtin MFMAs (the
    // builtin version of this loop spilled and shuffled 3 400 accvgpr copies: 1.27 ms per launch) nor counts asm reads.
    f32x16 Y[kRlNT];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    [&]<int... T>(std::integer_sequence<int, T...>) { ((Y[T] = zero16, rl_pin<T>(Y[T])), ...); }(std::make_integer_sequence<int, kRlNT>{});

    // fragment read of the rows: lane (r32, h) wants chunk 2 s + h of row 32 wave + r32 (s = k-step within the group)
    const int arow = 32 * wave + r32;
    const unsigned afr = rl_lds(abuf) + arow * 128;
    const int asw = (arow >> 1) & 7;
    const unsigned wlo = rl_lds(ring) + lane * 16, whi = wlo + 65536u;   // ds offsets are 16 bits

    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // A groups 0, 1 and the weights of k-steps 0, 1, 2 have landed
    constexpr int PD = 8;                                                // fragment reads in flight ahead of their MFMA
    bf16x8 wa[PD], xf[2];
    rl_read<0>(xf[0], afr + ((h ^ asw) << 4));                          // k-step 0 (A buffer 0): chunk h of the row
    [&]<int... J>(std::integer_sequence<int, J...>) { (rl_read<J * 1024>(wa[J], wlo), ...); }(std::make_integer_sequence<int, PD>{});
    for (int g = 0; g < NG; ++g) {
        const bool more_a = g + 2 < NG;
        const unsigned abase = afr + (unsigned)(g % 3) * kRlABuf, anext = afr + (unsigned)((g + 1) % 3) * kRlABuf;
        [&]<int... GI>(std::integer_sequence<int, GI...>) {
            ([&] {
                constexpr int gi = GI, i = gi / kRlNT, t = gi % kRlNT;     // k-step i of the group (ring slot i: KS % 4 == 0), output tile t
                const int ks = 4 * g + i;
                if constexpr (t == 2) {
                    // early in the k-step: the weights of the NEXT k-step (requested during the k-step before last) have landed -- younger than
                    // their last piece are the k-step behind them (6 pieces) and the rows' operand pieces around them (i == 1: the 4 of the
                    // k-step before; i == 2: the one behind that last piece); every wave is past the k-step before, whose slot the requests
                    // of k-step + 3 may now overwrite
                    if (ks + 2 >= KS) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    else if (i == 1 && more_a) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
                    else if (i == 2 && more_a) asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
                }
                if constexpr (t >= 3 && t <= 18 && t % 3 == 0) { if (ks + 3 < KS) dma_w1(ks + 3, (t - 3) / 3); }
                if constexpr (i == 0 && (t == 4 || t == 10 || t == 16 || t == 22)) { if (more_a) dma_a1(g + 2, (t - 4) / 6); }
                if constexpr (t == kRlNT / 2) {
                    // the rows' fragment of the next k-step (its group landed long ago: requested 8 k-steps before its first use)
                    const unsigned na = (i < 3 ? abase : anext) + ((((2 * ((i + 1) & 3)) + h) ^ asw) << 4);
                    rl_read<0>(xf[(i + 1) & 1], na);
                }
                // fragment PD gaps ahead: this k-step's slot, or the next k-step's (confirmed at the barrier above: t + PD >= 24 > 2)
                constexpr int tn = t + PD, sl = tn < kRlNT ? i : ((i + 1) & 3), tt = tn < kRlNT ? tn : tn - kRlNT;
                constexpr int LO = sl * kRlBlk + tt * 1024;
                rl_gap<PD - 1, (LO < 65536 ? LO : LO - 65536), (t < kRlNA)>(Y[t], wa[gi % PD], xf[i & 1], LO < 65536 ? wlo : whi);
            }(), ...);
        }(std::make_integer_sequence<int, 4 * kRlNT>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // the run-ahead fragment reads; the MFMA pipe drains before hipcc's reads of Y
    [&]<int... T>(std::integer_sequence<int, T...>) { (rl_pin<T>(Y[T]), ...); }(std::make_integer_sequence<int, kRlNT>{});

    // ---- epilogue: lane = row, register quad q of tile t = columns 32 t + 8 q + 4 h .. + 3
    // (a k-slice of the extra-token rows: the partial sums alone go to this workgroup's slab -- mlp_reduce_kernel adds bias, x and the slabs in
    // a fixed order; the same code path with the slab as the destination, so that the 384 accumulators are read in one place)
    // The residual rows come in by LDS-DMA, 256 columns of the wave's 32 rows at a time (32 requests = 32 KB in flight per wave): the ring is free now,
    // and register loads -- a handful in flight next to 384 accumulator registers -- left the epilogue latency-bound (190 of the fc2 launch's 455 us).
    asm volatile("s_barrier" ::: "memory");          // every wave's fragment reads of the ring are done
    constexpr int kPitch = 1024 + 16;                 // lane = row reads of a column quad: 16 consecutive rows fall into 16 different bank groups
    char* strip = smem + wave * (32 * kPitch);
    bool ok;
    const long long rr = row_of(arow, ok);
    bool ok0;
    const long long row0w = __builtin_amdgcn_readfirstlane((int)row_of(32 * wave, ok0));      // the wave's 32 rows are consecutive token rows (tok_n % 32 == 0)
    const int limit = !planned ? a.M : a.n_main;
    const long long rows_all = planned ? (long long)(a.n_main / a.tok_n) * a.tok_l : a.M;
    int nvalid = limit - (tile * 128 + 32 * wave);
    nvalid = part_tile ? 32 : (nvalid < 0 ? 0 : (nvalid > 32 ? 32 : nvalid));
    // (straight-line: residual and bias enter through a 0 / 1 factor -- exact either way -- a branch or select here makes hipcc spill 350 registers)
    const float keep = part_tile || a.set_x ? 0.f : 1.f, keepb = part_tile ? 0.f : 1.f;      // residual / bias factors
    ok = ok || part_tile;
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
    float cshift = 0.f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (a.set_x || part_tile) {   // no residual (skip_linear; a slab tile: the reduce launch adds x): the strip (leftover ring bytes) is zeroed so that the 0 factor meets finite values
#pragma unroll
            for (int j = 0; j < 32; ++j) *reinterpret_cast<f32x4*>(strip + r32 * kPitch + (h * 32 + j) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const char* src = reinterpret_cast<const char*>(a.xres + 256 * p);
            for (int i = 0; i < nvalid; ++i) {
                const long long ri = row0w + i < rows_all ? row0w + i : rows_all - 1;
                rl_dma(src + ri * (kRlD * 4), lane * 16, strip + i * kPitch);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            const int t = 8 * p + tt;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4* sl = reinterpret_cast<f32x4*>(strip + r32 * kPitch + (32 * tt + 8 * q + 4 * h) * 4);
                const f32x4 xv = *sl;
                const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + 32 * t + 8 * q + 4 * h);
                f32x4 v = {Y[t][4 * q], Y[t][4 * q + 1], Y[t][4 * q + 2], Y[t][4 * q + 3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(xv[e], keep, v[e] + bv[e] * keepb);
                *sl = v;
                if (t == 0 && q == 0) cshift = v[0];
                const f32x4 d = v - cshift;
                s4 += d;
                q4 += d * d;
#pragma unroll
                for (int e = 0; e < 4; ++e) Y[t][4 * q + e] = v[e];
            }
        }
        // the updated rows leave as whole 1 KB segments (lane = 16-byte column chunk), the bf16 copy as 512-byte segments
        {
            float* dst = (part_tile ? a.partial + ((long long)xe * 128 + 32 * wave) * kRlD : a.xres + row0w * kRlD) + 256 * p + 4 * lane;
            bf16_t* cp = a.x_copy && !part_tile ? a.x_copy + row0w * kRlD + 256 * p + 4 * lane : nullptr;
#pragma unroll 4
            for (int i = 0; i < 32; ++i) {
                if (i < nvalid) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(strip + i * kPitch + lane * 16);
                    *reinterpret_cast<f32x4*>(dst + (long long)i * kRlD) = v;
                    if (cp) *reinterpret_cast<uint2*>(cp + (long long)i * kRlD) = uint2{rl_pack2(v[0], v[1]), rl_pack2(v[2], v[3])};
                }
            }
        }
    }
    if ((a.h_out || a.h_frag) && !part_tile) {
        // one-pass shifted statistics of the row (its two lane halves combined exactly), as the fused block tail's LayerNorms
        constexpr float n = (float)(kRlD / 2);
        const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]), qq = (q4[0] + q4[1]) + (q4[2] + q4[3]);
        const float mh = cshift + s / n, m2h = qq - s * s / n;
        const auto pm = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mh), __builtin_bit_cast(unsigned, mh), false, false);
        const auto p2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2h), __builtin_bit_cast(unsigned, m2h), false, false);
        const unsigned u_lo = pm[0], u_hi = pm[1], w_lo = p2[0], w_hi = p2[1];
        const float m_lo = __builtin_bit_cast(float, u_lo), m_hi = __builtin_bit_cast(float, u_hi);
        const float q_lo = __builtin_bit_cast(float, w_lo), q_hi = __builtin_bit_cast(float, w_hi);
        const float mean = 0.5f * (m_lo + m_hi);
        const float dm = m_lo - m_hi;
        const float var = ((q_lo + q_hi) + dm * dm * (0.5f * n)) / (float)kRlD;
        const float rstd = 1.0f / sqrtf((var > 0.f ? var : 0.f) + 1e-5f);
        const float shift = -mean * rstd;
        // row-major rows, or (h_frag: main tiles) the MFMA fragment order the attention launch loads: [32-row group][k-step][lane] x 16 bytes --
        // the 16-byte piece of (tile t, quad pair qp) is k-step 2 t + qp / 2 of this wave's group, at this lane's slot
        // fragment order: whole 1 KB pieces per request as the registers stand.  Row-major (the next GEMM's operand): turned through the strip, 384
        // columns at a time, so that a request stores 768 contiguous bytes of one row instead of 32 rows x 16 bytes
        constexpr int kHPitch = 768 + 16;
        bf16_t* hr = a.h_frag ? a.h_frag + (((long long)tile * 4 + wave) * (kRlD / 16) * 64 + lane) * 8 : nullptr;
#pragma unroll
        for (int t = 0; t < kRlNT; ++t) {
            uint2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 gv = *reinterpret_cast<const f32x4*>(a.ln_g + 32 * t + 8 * q + 4 * h), bv = *reinterpret_cast<const f32x4*>(a.ln_b + 32 * t + 8 * q + 4 * h);
                const f32x4 yv = {Y[t][4 * q], Y[t][4 * q + 1], Y[t][4 * q + 2], Y[t][4 * q + 3]};
                const f32x4 w = (yv * rstd + shift) * gv + bv;
                v[q] = uint2{rl_pack2(w[0], w[1]), rl_pack2(w[2], w[3])};
            }
#pragma unroll
            for (int qp = 0; qp < 4; qp += 2) {      // 16-byte row segments (v_permlane32_swap pairs the lane halves)
                const auto s0 = __builtin_amdgcn_permlane32_swap(v[qp].x, v[qp + 1].x, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(v[qp].y, v[qp + 1].y, false, false);
                const uint4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (hr) { if (ok) *reinterpret_cast<uint4*>(hr + (2 * t + qp / 2) * 512) = o; }
                else *reinterpret_cast<uint4*>(strip + r32 * kHPitch + (32 * (t % 12) + 8 * qp + 8 * h) * 2) = o;
            }
            if (!hr && t % 12 == 11) {
                bf16_t* dst = a.h_out + row0w * kRlD + 384 * (t / 12) + 8 * lane;
#pragma unroll 4
                for (int i = 0; i < 32; ++i)
                    if (i < nvalid && lane < 48) *reinterpret_cast<uint4*>(dst + (long long)i * kRlD) = *reinterpret_cast<const uint4*>(strip + i * kHPitch + lane * 16);
            }
        }
    }
}

}  // namespace

bool rowlin_supported(int D, int K) { return D == kRlD && K % 64 == 0 && K >= 192; }

// nn.Linear weight [768, K] -> the stream the kernel reads: [k-step][output tile t][lane] x 16 bytes, each 1 KB fragment in MFMA A-operand
// order: img[((ks * 24 + t) * 64 + lane) * 8 + i] = W[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + i]
void rowlin_pack(int K, const float* w, unsigned short (*to_bf16)(float), unsigned short* img) {
    for (int ks = 0; ks < K / 16; ++ks)
        for (int t = 0; t < kRlNT; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                const float* src = w + (size_t)(32 * t + (lane & 31)) * K + 16 * ks + 8 * (lane >> 5);
                unsigned short* dst = img + (((size_t)ks * kRlNT + t) * 64 + lane) * 8;
                for (int i = 0; i < 8; ++i) dst[i] = to_bf16(src[i]);
            }
}

// Row plan: patch rows in 128-row main tiles, extra-token rows in 128-row tiles split `groups` ways along K (a function of K alone -- never of
// the batch -- so results do not depend on the batch size)
void rowlin_plan(int B, int n_patches, int extras, int seq_len, int K, RowLinArgs& a) {
    a.tok_n = n_patches; a.tok_e = extras; a.tok_l = seq_len;
    a.n_main = B * n_patches; a.n_extra = B * extras;
    a.tiles_main = (a.n_main + 127) / 128;
    a.tiles_extra = (a.n_extra + 127) / 128;
    const int ks = K / 16;
    int g = 16;
    while (g > 1 && (ks / g < 4 || (ks / g) % 4 || ks % g)) g >>= 1;     // k-steps per group: a multiple of 4, the same for every group
    a.groups = g;
    a.cpg = ks / g;
}
size_t rowlin_partial_bytes(int max_batch, int extras, int K) {
    RowLinArgs a{};
    rowlin_plan(max_batch, 32, extras, 32 + extras, K, a);
    return (size_t)a.tiles_extra * a.groups * 128 * kRlD * sizeof(float);
}

hipError_t launch_rowlin(const RowLinArgs& a, hipStream_t s) {
    if (!rowlin_supported(kRlD, a.K) || (a.tok_n <= 0 && a.M < 1) || !a.A || !a.wimg || !a.bias || !a.xres) return hipErrorInvalidValue;
    if (a.k_split && (a.k_split % 64 || a.k_split >= a.K || !a.A2 || (a.tok_n > 0 && (a.cpg * 16) % 64))) return hipErrorInvalidValue;
    const long long rows_all = a.tok_n > 0 ? (long long)(a.n_main / a.tok_n) * a.tok_l : a.M;
    if (rows_all * a.lda * 2 >= (1ll << 32)) return hipErrorInvalidValue;     // the kernel's operand row offsets are 32 bits
    const int grid = a.tok_n > 0 ? a.tiles_main + a.tiles_extra * a.groups : (a.M + 127) / 128;
    if (a.tok_n > 0 && (a.cpg % 4 || a.groups < 1 || (a.tiles_extra > 0 && !a.partial) || a.tok_n % 32)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(rowlin768_kernel, dim3(grid), dim3(256), kRlLds, s, a);
    return hipGetLastError();
}

hipError_t init_rowlin_kernels() {
    return hipFuncSetAttribute((const void*)rowlin768_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kRlLds);
}

}  // namespace dd
