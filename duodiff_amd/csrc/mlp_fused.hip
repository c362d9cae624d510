// Fused U-ViT MLP for gfx950:  x += fc2( GELU_erf( fc1(h) + b1 ) ) + b2   in ONE launch, bf16 MFMA operands,
// fp32 accumulation and fp32 residual stream.  Replaces reference models/uvit.py:86-92 (Mlp.forward) plus the
// residual add of Block.forward (models/uvit.py:207).  The hidden activation [M, 4D] (135 MB per block at B = 128)
// never exists: not in HBM, not in LDS.
//
// Shape of the computation (D = embedding dim <= 512, hidden = 4D, both multiples of 32):
//   * a workgroup owns 128 token rows, a wave 32 of them -- for ALL D output columns, so one wave carries its
//     whole output tile Y[32, D] in accumulator registers (D/2 registers per lane; 256 at D = 512: the kernel runs
//     one wave per SIMD with the full 512-entry register file) and its input rows X[32, D] as MFMA B-operand
//     fragments (D/4 registers per lane), loaded once;
//   * the hidden dimension is walked in chunks of 32:  S^T[32 hidden, 32 rows] = W1_c . X^T  (v_mfma_f32_32x32x16_bf16,
//     weights as the A operand, so the LANE is the token row and the 16 accumulator registers are hidden units),
//     GELU in registers, and the bf16-packed accumulator IS the B operand of  Y^T += W2_c . P^T  -- its k order is the
//     accumulator row order (e&3) + 8(e>>2) + 4(lane>>5); W2 is stored with its k index permuted to match on the host;
//   * weights stream through LDS as a ring of 4 blocks (W1 chunk / W2 chunk alternating), LDS-DMA'd from an image that
//     is already in MFMA fragment order (1 KB per fragment = 64 lanes x 16 B): the DMA is linear, every fragment read
//     is a conflict-free ds_read_b128 at a compile-time offset from one address register;
//   * schedule per chunk c (three chunks in flight): [GEMM2 of chunk c-1] then [GEMM1 of chunk c+1], with the GELU of
//     chunk c spread over the gaps behind all 2F MFMAs as two interleaved dependency chains (3 VALU instructions per
//     gap); every block is requested 1.5 chunk-times before its first use (counted vmcnt, two raw barriers per chunk).
// Rows.  The token matrix holds per image `extras` (1-2) time / label tokens followed by N patch tokens.  The PATCH rows of
// all images form the main row space, cut into tiles of 128: B * 256 patch rows = 2B tiles, exactly one round over the
// 256 CUs at the headline batch of 128 (a tile of 128 consecutive token rows would give 257 tiles: a second round for
// one tile).  The few EXTRA rows (B * extras) are gathered into tiles of their own, each split along the hidden dimension
// over a fixed number of workgroups that write partial sums to slabs; a small reduce kernel folds the slabs into x in a
// fixed order (deterministic, no atomics).  Which path a row takes depends only on its token index, never on the batch
// size, so an image computes bit-identically alone and inside any batch.
#include "dd_internal.h"

#include <type_traits>
#include <utility>

namespace dd {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack2(float lo, float hi) {   // v_cvt_pk_bf16_f32
    typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
    typedef float f32v2 __attribute__((ext_vector_type(2)));
    const f32v2 q = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16v2));
}

// The hot loop is written as a sequence of small asm volatile statements -- one MFMA, one LDS fragment read, or one half
// of a GELU evaluation each.  Volatile asm statements keep their program order, so the interleave written in the source
// IS the instruction stream (hipcc sinks or hoists plain C++ VALU code around asm MFMAs as it likes, and drains
// lgkmcnt to 0 in front of every asm that consumes one of its own ds_reads).  The price: hipcc neither counts the
// asm ds_reads nor pads hazards around asm, so
//   * every MFMA statement carries its own counted s_waitcnt lgkmcnt(N): LDS reads return in order, the fragment
//     queue is PD deep, so "at most PD-1 outstanding" means the fragment of this MFMA has landed;
//   * an MFMA whose B operand was just written by VALU (v_cvt_pk_bf16_f32) starts with s_nop 1;
//   * an MFMA result is not read by VALU code for 12 wait states (mfma_drain) unless a long MFMA sequence intervenes.
// Accumulators: Y tiles "+a" (AGPR half of the register file: 256 registers at D = 512, resident for the whole kernel),
// hidden-chunk accumulators "+v" (the GELU reads them with VALU instructions).
template <int LGKM, bool NOP>
__device__ __forceinline__ void mfma_acc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    if constexpr (NOP)
        asm volatile("s_waitcnt lgkmcnt(%3)\n\ts_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b), "i"(LGKM));
    else
        asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b), "i"(LGKM));
}
template <int LGKM>
__device__ __forceinline__ void mfma_hid(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b), "i"(LGKM));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ void lds_frag(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_quad(f32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

// exact-erf GELU for a bf16-rounded result: gelu(v) = hv + hv * s * P(s^2), s = med3(v, +-3.8), hv = v/2, P the degree-6
// minimax polynomial of erf(s/sqrt2)/s (same coefficients as the GEMM epilogue in gemm.hip: |erf error| <= 1.3e-4, GELU
// abs error <= 2.4e-4).  TWO values are evaluated together as interleaved chains (no instruction reads the result of the
// one in front of it), cut into 8 pieces of 3 VALU instructions; one piece sits in the gap behind one MFMA.
struct GeluConst { float hi, c5; };   // 3.8 and the s^12 coefficient live in VGPRs (VOP3 / fmamk take no second literal)
struct GeluPair { float sa, s2a, pa, ha, sb, s2b, pb, hb; };
template <int K>
__device__ __forceinline__ void gelu_piece(float va, float vb, const GeluConst& k, GeluPair& r, unsigned& out) {
    if constexpr (K == 0)
        asm volatile("v_med3_f32 %0, %3, %5, %6\n\tv_med3_f32 %1, %4, %5, %6\n\tv_mul_f32 %2, %0, %0"
                     : "=&v"(r.sa), "=&v"(r.sb), "=&v"(r.s2a) : "v"(va), "v"(vb), "s"(-3.8f), "v"(k.hi));
    else if constexpr (K == 1)     //  7.331517960e-08 * s2 + c5
        asm volatile("v_mul_f32 %0, %3, %3\n\tv_fmamk_f32 %1, %4, 0x339d7172, %5\n\tv_fmamk_f32 %2, %0, 0x339d7172, %5"
                     : "=&v"(r.s2b), "=&v"(r.pa), "=&v"(r.pb) : "v"(r.sb), "v"(r.s2a), "v"(k.c5));
    else if constexpr (K == 2)     //  * s2 + 1.213693460e-04 ; a: * s2 - 1.863093246e-03
        asm volatile("v_fmaak_f32 %0, %0, %2, 0x38fe87ac\n\tv_fmaak_f32 %1, %1, %3, 0x38fe87ac\n\tv_fmaak_f32 %0, %0, %2, 0xbaf43309"
                     : "+v"(r.pa), "+v"(r.pb) : "v"(r.s2a), "v"(r.s2b));
    else if constexpr (K == 3)     //  b: * s2 - 1.863093246e-03 ; * s2 + 1.863326334e-02
        asm volatile("v_fmaak_f32 %1, %1, %3, 0xbaf43309\n\tv_fmaak_f32 %0, %0, %2, 0x3c98a4c9\n\tv_fmaak_f32 %1, %1, %3, 0x3c98a4c9"
                     : "+v"(r.pa), "+v"(r.pb) : "v"(r.s2a), "v"(r.s2b));
    else if constexpr (K == 4)     //  * s2 - 1.314395642e-01 ; a: * s2 + 7.973534865e-01
        asm volatile("v_fmaak_f32 %0, %0, %2, 0xbe069818\n\tv_fmaak_f32 %1, %1, %3, 0xbe069818\n\tv_fmaak_f32 %0, %0, %2, 0x3f4c1f5c"
                     : "+v"(r.pa), "+v"(r.pb) : "v"(r.s2a), "v"(r.s2b));
    else if constexpr (K == 5)     //  b: * s2 + 7.973534865e-01 ; e = P * s
        asm volatile("v_fmaak_f32 %1, %1, %2, 0x3f4c1f5c\n\tv_mul_f32 %0, %0, %3\n\tv_mul_f32 %1, %1, %4"
                     : "+v"(r.pa), "+v"(r.pb) : "v"(r.s2b), "v"(r.sa), "v"(r.sb));
    else if constexpr (K == 6)     //  hv ; a: hv + hv * e
        asm volatile("v_mul_f32 %0, 0.5, %2\n\tv_mul_f32 %1, 0.5, %3\n\tv_fmac_f32 %0, %0, %4"
                     : "=&v"(r.ha), "=&v"(r.hb) : "v"(va), "v"(vb), "v"(r.pa));
    else                           //  b: hv + hv * e ; pack the pair
        asm volatile("v_fmac_f32 %1, %1, %3\n\tv_cvt_pk_bf16_f32 %0, %2, %1"
                     : "=&v"(out), "+v"(r.hb) : "v"(r.ha), "v"(r.pb));
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <int D>
struct MlpCfg {
    static constexpr int NT = D / 32;          // 32-column output tiles of one wave
    static constexpr int KS = D / 16;          // k-steps (16 wide) of the first GEMM
    static constexpr int F = D / 16;           // 1 KB fragments per ring block (W1 chunk: KS, W2 chunk: 2 * NT)
    static constexpr int BLK = F * 1024;       // bytes per ring block
    static constexpr int RING = 4 * BLK;
    static constexpr int FPW = F / 4;          // fragments each of the 4 waves DMAs per block
};

template <int D>
__global__ void __launch_bounds__(256) mlp_fused_kernel(const MlpFusedArgs a) {
    using C = MlpCfg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* b1s = reinterpret_cast<float*>(smem + C::RING);      // [hidden], in accumulator-register order per chunk

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;

    // ---- which rows, which hidden chunks
    int tile, c0, c1, slab = 0;
    bool partial = false;
    if ((int)blockIdx.x < a.tiles_main) {
        tile = blockIdx.x; c0 = 0; c1 = a.nchunks;
    } else {
        const int e = blockIdx.x - a.tiles_main;
        const int lt = e / a.groups, g = e - lt * a.groups;
        tile = a.tiles_main + lt;
        c0 = g * a.cpg;
        c1 = c0 + a.cpg < a.nchunks ? c0 + a.cpg : a.nchunks;
        slab = e;
        partial = true;
    }
    // logical row of this lane -> physical token row (clamped to a valid row; stores are masked by row_ok)
    long long row;
    bool row_ok;
    {
        const int idx = (partial ? tile - a.tiles_main : tile) * 128 + wave * 32 + r32;
        if (!partial) {
            row_ok = idx < a.n_main;
            const int p = row_ok ? idx : 0, b = p / a.tok_n;
            row = (long long)b * a.tok_l + a.tok_e + (p - b * a.tok_n);
        } else {
            row_ok = idx < a.n_extra;
            const int q = row_ok ? idx : 0, b = q / a.tok_e;
            row = (long long)b * a.tok_l + (q - b * a.tok_e);
        }
    }

    // ---- LDS-DMA of one ring block.  Stream position b: chunk b>>1, W1 block if b is even, W2 block if odd.
    auto dma_block = [&](int b, int slot) {
        const char* src = a.wimg + (size_t)b * C::BLK + (wave * C::FPW) * 1024 + lane * 16;
        char* dst = smem + slot * C::BLK + (wave * C::FPW) * 1024;
#pragma unroll
        for (int j = 0; j < C::FPW; ++j) glds16(src + j * 1024, dst + j * 1024);
    };

    // prologue: W1(c0), W2(c0), W1(c0+1) in flight while the X fragments and the bias table are fetched.  c0 is even
    // (mlp_fused_plan), so block b always lives in ring slot b & 3: W1(c) in slot 0 / 2, W2(c) in slot 1 / 3.
    dma_block(2 * c0, 0);
    dma_block(2 * c0 + 1, 1);
    dma_block(2 * c0 + 2, 2);
    {   // slot 3 stands in for "W2 of chunk c0-1": zeros, so that the first iteration's GEMM2 adds nothing
        f32x4* z = reinterpret_cast<f32x4*>(smem + 3 * C::BLK);
        for (int i = tid; i < C::BLK / 16; i += 256) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    bf16x8 xf[C::KS];
    {
        const bf16_t* xr = a.X + row * a.ldx + 8 * h;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
    }
    for (int i = tid; i < a.nchunks * 8; i += 256)
        reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(a.b1p)[i];

    f32x16 Y[C::NT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) Y[t][e] = 0.f;

    // LDS addressing: one per-lane base (+ a second one 64 KB up: ds offsets are 16 bits), compile-time offsets
    const unsigned lds_lo = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
    const unsigned lds_hi = lds_lo + 65536u;
    const unsigned bias_lo = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + C::RING + h * 64;
    constexpr int PD = C::F < 8 ? C::F : 8;    // fragment reads in flight ahead of their MFMA: covers ~250 cycles of LDS latency
    const GeluConst gk{3.8f, -4.544908101e-06f};

    // fragment F of the slot's block -> register q; slot and fragment are compile-time, so this is one ds_read_b128
    auto frag = [&](auto slot_tag, auto f_tag, bf16x8& q) {
        constexpr int OFF = decltype(slot_tag)::value * C::BLK + decltype(f_tag)::value * 1024;
        if constexpr (OFF < 65536) lds_frag<OFF>(q, lds_lo);
        else lds_frag<OFF - 65536>(q, lds_hi);
    };
    // S accumulator of chunk c initialised with its fc1 bias (register e of lane half h = hidden 32c + (e&3) + 8(e>>2) + 4h)
    auto bias_init = [&](int c, f32x16& sacc) {
        f32x4 q0, q1, q2, q3;
        const unsigned ba = bias_lo + c * 128;
        lds_quad<0>(q0, ba); lds_quad<16>(q1, ba); lds_quad<32>(q2, ba); lds_quad<48>(q3, ba);
        sacc = f32x16{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3],
                      q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
    };
    // the F MFMAs of one phase, fed by a fragment queue PD deep: MFMA f, the read of fragment f + PD, then `gap(f)`
    auto phase = [&](auto slot_tag, bf16x8 (&wq)[PD], auto&& mma, auto&& gap) {
        [&]<int... J>(std::integer_sequence<int, J...>) { (frag(slot_tag, std::integral_constant<int, J>{}, wq[J]), ...); }
        (std::make_integer_sequence<int, PD>{});
        [&]<int... FI>(std::integer_sequence<int, FI...>) {
            ([&] {
                constexpr int f = FI, left = C::F - 1 - f;
                mma(std::integral_constant<int, f>{}, std::integral_constant<int, (left < PD - 1 ? left : PD - 1)>{}, wq[f % PD]);
                if constexpr (f + PD < C::F) frag(slot_tag, std::integral_constant<int, f + PD>{}, wq[f % PD]);
                gap(std::integral_constant<int, f>{});
            }(), ...);
        }(std::make_integer_sequence<int, C::F>{});
    };

    __syncthreads();   // (vmcnt(0): the prologue blocks have landed; bias table and zero block are visible)

    f32x16 sA, sB;
    bf16x8 wq[PD];
    bias_init(c0, sA);
    phase(std::integral_constant<int, 0>{}, wq,
          [&](auto f, auto lg, const bf16x8& w) { mfma_hid<decltype(lg)::value>(sA, w, xf[decltype(f)::value]); },
          [&](auto) {});

    // One chunk c, PAR = c & 1.  s_cur = S of chunk c (complete), s_next receives S of chunk c+1, p_prev = GELU of chunk
    // c-1 (zeros for the first chunk), p_out receives the GELU of chunk c.  No branch inside: the last chunk computes a
    // throw-away S from the padded image.
    //   E: wait W2(c-1) landed | barrier | request W1(c+2) into the slot W1(c) has left
    //      phase A  Y += W2(c-1) . p_prev                      gaps: GELU pieces 0 .. F*PPG-1 of chunk c
    //   M: wait W1(c+1) landed | barrier | request W2(c+1) into the slot W2(c-1) has left
    //      phase B  s_next = b1(c+1) + W1(c+1) . X             gaps: the remaining GELU pieces
    // DMA groups retire in issue order and at most three are outstanding, so "at most 2 * FPW instructions outstanding"
    // (counted vmcnt) is "the oldest group has landed"; the raw barrier then extends that to every wave's pieces and
    // orders the slot hand-over.  Nothing else in the loop touches vmcnt.
    constexpr int PPG = 32 / C::F;             // GELU pieces per gap: 64 pieces (8 pairs x 8) over 2F gaps
    auto iteration = [&](auto par_tag, int c, f32x16& s_cur, f32x16& s_next, const bf16x8 (&p_prev)[2], bf16x8 (&p_out)[2]) {
        constexpr int PAR = decltype(par_tag)::value;
        using SlotW2P [[maybe_unused]] = std::integral_constant<int, PAR ? 1 : 3>;      // W2 of chunk c-1: block 2c-1
        using SlotW1N [[maybe_unused]] = std::integral_constant<int, PAR ? 0 : 2>;      // W1 of chunk c+1: block 2c+2
        GeluPair gr;
        unsigned pw[8];
        [[maybe_unused]] auto gap = [&](auto g_tag) {           // gap g of the iteration (0 .. 2F-1): PPG pieces
            [&]<int... Q>(std::integer_sequence<int, Q...>) {
                ([&] {
                    constexpr int id = decltype(g_tag)::value * PPG + Q, pair = id >> 3;
                    gelu_piece<(id & 7)>(s_cur[2 * pair], s_cur[2 * pair + 1], gk, gr, pw[pair]);
                }(), ...);
            }(std::make_integer_sequence<int, PPG>{});
        };
        wait_vmcnt<2 * C::FPW>();
        __builtin_amdgcn_s_barrier();
#if !defined(DD_MLP_ABLATE) || DD_MLP_ABLATE != 1     // development builds only (tools/build_variant.py): 1 = no DMA in the loop
        dma_block(2 * c + 4, PAR ? 2 : 0);
#endif
#if !defined(DD_MLP_ABLATE) || DD_MLP_ABLATE != 2     // 2 = DMA stream + barriers only
        phase(SlotW2P{}, wq,
              [&](auto f, auto lg, const bf16x8& w) {
                  constexpr int fi = decltype(f)::value;
                  mfma_acc<decltype(lg)::value, false>(Y[fi >> 1], w, p_prev[fi & 1]);
              },
              [&](auto f) { gap(f); });
#endif
        wait_vmcnt<2 * C::FPW>();
        __builtin_amdgcn_s_barrier();
#if !defined(DD_MLP_ABLATE) || DD_MLP_ABLATE != 1
        dma_block(2 * c + 3, PAR ? 1 : 3);
#endif
#if !defined(DD_MLP_ABLATE) || DD_MLP_ABLATE != 2
        bias_init(c + 1, s_next);
        phase(SlotW1N{}, wq,
              [&](auto f, auto lg, const bf16x8& w) { mfma_hid<decltype(lg)::value>(s_next, w, xf[decltype(f)::value]); },
              [&](auto f) { gap(std::integral_constant<int, C::F + decltype(f)::value>{}); });
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        p_out[0] = __builtin_bit_cast(bf16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
        p_out[1] = __builtin_bit_cast(bf16x8, u32x4{pw[4], pw[5], pw[6], pw[7]});
#endif
    };

    bf16x8 pA[2], pB[2];
    pA[0] = pA[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = c0; c < c1; c += 2) {   // c0 and c1 are even (mlp_fused_plan): no control flow around the accumulators
        iteration(std::integral_constant<int, 0>{}, c, sA, sB, pA, pB);
        iteration(std::integral_constant<int, 1>{}, c + 1, sB, sA, pB, pA);
    }
    // tail: GEMM2 of the last chunk (c1 - 1 is odd: its W2 block sits in slot 3)
    wait_vmcnt<2 * C::FPW>();
    __builtin_amdgcn_s_barrier();
    phase(std::integral_constant<int, 3>{}, wq,
          [&](auto f, auto lg, const bf16x8& w) {
              constexpr int fi = decltype(f)::value;
              mfma_acc<decltype(lg)::value, (fi < 2)>(Y[fi >> 1], w, pA[fi & 1]);
          },
          [&](auto) {});
    wait_vmcnt<0>();     // the run-ahead DMA of the padded blocks must not outlive the workgroup's LDS allocation
    // hipcc does not know that the asm statements are MFMAs: left alone it schedules its own reads of the accumulators
    // (v_accvgpr_read, AGPR spills) directly behind the last MFMA, inside its 12-wait-state shadow.  The drain, then one
    // empty asm per tile that "rewrites" it: every compiler read of Y is ordered behind the drain.
    mfma_drain();
#pragma unroll
    for (int t = 0; t < C::NT; ++t) asm volatile("" : "+a"(Y[t]));

    // ---- epilogue.  Accumulator layout: lane = token row, register quad g of tile t = columns 32t + 8g + 4h .. +3.
    if (partial) {
        float* pp = a.partial + ((long long)slab * 128 + wave * 32 + r32) * D + 4 * h;
#pragma unroll
        for (int t = 0; t < C::NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(pp + 32 * t + 8 * g) = f32x4{Y[t][4 * g], Y[t][4 * g + 1], Y[t][4 * g + 2], Y[t][4 * g + 3]};
        return;
    }
    float* xrow = a.xres + row * D + 4 * h;
    const float* b2 = a.b2 + 4 * h;
    f32x4 xl[2][4];   // residual quads of tile t, loaded one tile ahead of the stores (vmcnt retires in order)
#pragma unroll
    for (int g = 0; g < 4; ++g) xl[0][g] = *reinterpret_cast<const f32x4*>(xrow + 8 * g);
#pragma unroll
    for (int t = 0; t < C::NT; ++t) {
        if (t + 1 < C::NT) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xl[(t + 1) & 1][g] = *reinterpret_cast<const f32x4*>(xrow + 32 * (t + 1) + 8 * g);
        }
        uint2 v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 q = {Y[t][4 * g], Y[t][4 * g + 1], Y[t][4 * g + 2], Y[t][4 * g + 3]};
            q += *reinterpret_cast<const f32x4*>(b2 + 32 * t + 8 * g);
            q = xl[t & 1][g] + q;
            if (row_ok) *reinterpret_cast<f32x4*>(xrow + 32 * t + 8 * g) = q;
            v[g] = uint2{pack2(q[0], q[1]), pack2(q[2], q[3])};
        }
        if (a.out) {
            // bf16 copy as 16-byte row segments: v_permlane32_swap joins the two lane halves (see gemm.hip)
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(v[gp].x, v[gp + 1].x, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(v[gp].y, v[gp + 1].y, false, false);
                const uint4 o = {s0[0], s1[0], s0[1], s1[1]};
                if (row_ok) *reinterpret_cast<uint4*>(a.out + row * a.ldo + 32 * t + 8 * gp + 8 * h) = o;
            }
        }
    }
}

// x[row] += b2 + sum over the groups' partial slabs (fixed order), for the rows of the leftover tiles
__global__ void __launch_bounds__(256) mlp_reduce_kernel(const MlpFusedArgs a, int D) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;     // one float4 each
    const int qpr = D / 4;
    if (i >= (long long)a.n_extra * qpr) return;
    const long long r = i / qpr;                                       // logical extra row
    const int col = (int)(i - r * qpr) * 4;
    const int lt = (int)(r / 128), rr = (int)(r % 128);
    const long long b = r / a.tok_e;
    const long long row = b * a.tok_l + (r - b * a.tok_e);
    f32x4 acc = *reinterpret_cast<const f32x4*>(a.b2 + col);
    for (int g = 0; g < a.groups; ++g)
        acc += *reinterpret_cast<const f32x4*>(a.partial + (((long long)lt * a.groups + g) * 128 + rr) * D + col);
    f32x4* xp = reinterpret_cast<f32x4*>(a.xres + row * D + col);
    const f32x4 q = *xp + acc;
    *xp = q;
    if (a.out) *reinterpret_cast<uint2*>(a.out + row * a.ldo + col) = uint2{pack2(q[0], q[1]), pack2(q[2], q[3])};
}

template <int D>
hipError_t launch_d(const MlpFusedArgs& a, hipStream_t s) {
    const size_t lds = MlpCfg<D>::RING + (size_t)(a.nchunks + 1) * 32 * sizeof(float);   // + one chunk: bias_init(c1) is read, unused
    const int grid = a.tiles_main + a.tiles_left * a.groups;
    hipLaunchKernelGGL(mlp_fused_kernel<D>, dim3(grid), dim3(256), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && a.tiles_left > 0) {
        const long long n4 = (long long)a.n_extra * (D / 4);
        hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, a, D);
        e = hipGetLastError();
    }
    return e;
}

constexpr int kMaxHidden = 4096;   // bias table in LDS next to the 128 KB ring

}  // namespace

bool mlp_fused_supported(int D, int hidden) {
    return (D == 64 || D == 128 || D == 256 || D == 512) && hidden % 64 == 0 && hidden >= 64 && hidden <= kMaxHidden;
}

// + four blocks: the kernel's DMA runs up to three blocks past the last chunk (branch-free pipeline); never used as data
size_t mlp_fused_image_bytes(int D, int hidden) { return (size_t)(hidden / 32 + 2) * 2 * (D / 16) * 1024; }

// Row plan (see the header): patch rows in 128-row main tiles, extra rows in tiles split `groups` ways along hidden.
// The split is a function of the hidden size alone -- never of the batch -- so results do not depend on the batch size.
void mlp_fused_plan(int B, int n_patches, int extras, int seq_len, int hidden, MlpFusedArgs& a) {
    const int nchunks = hidden / 32;
    a.nchunks = nchunks;
    a.tok_n = n_patches; a.tok_e = extras; a.tok_l = seq_len;
    a.n_main = B * n_patches;
    a.n_extra = B * extras;
    a.tiles_main = (a.n_main + 127) / 128;
    a.tiles_left = (a.n_extra + 127) / 128;
    int g = nchunks / 2 < 16 ? nchunks / 2 : 16;                     // >= 2 chunks per group (the kernel unrolls by 2)
    if (g < 1) g = 1;
    a.cpg = ((nchunks + g - 1) / g + 1) & ~1;
    a.groups = (nchunks + a.cpg - 1) / a.cpg;
}

size_t mlp_fused_partial_bytes(int max_batch, int extras, int D, int hidden) {
    MlpFusedArgs a{};
    mlp_fused_plan(max_batch, 1, extras, 1 + extras, hidden, a);
    return (size_t)a.tiles_left * a.groups * 128 * D * sizeof(float);
}

// Host: nn.Linear weights (fp32, [out, in]) -> the fragment-ordered bf16 image the kernel streams + permuted fc1 bias.
void mlp_fused_pack(int D, int hidden, const float* w1, const float* b1, const float* w2,
                    unsigned short (*to_bf16)(float), unsigned short* img, float* b1p) {
    const int F = D / 16, NT = D / 32, KS = D / 16, nchunks = hidden / 32;
    for (int c = 0; c < nchunks; ++c) {
        unsigned short* blk1 = img + (size_t)(2 * c) * F * 512;
        unsigned short* blk2 = img + (size_t)(2 * c + 1) * F * 512;
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 31, h = lane >> 5;
            for (int ks = 0; ks < KS; ++ks)
                for (int j = 0; j < 8; ++j)
                    blk1[(size_t)ks * 512 + lane * 8 + j] = to_bf16(w1[(size_t)(32 * c + r) * D + 16 * ks + 8 * h + j]);
            for (int t = 0; t < NT; ++t)
                for (int s = 0; s < 2; ++s)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 32 * c + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);   // accumulator row order
                        blk2[(size_t)(2 * t + s) * 512 + lane * 8 + j] = to_bf16(w2[(size_t)(32 * t + r) * hidden + k]);
                    }
        }
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 16; ++e) b1p[c * 32 + h * 16 + e] = b1[32 * c + (e & 3) + 8 * (e >> 2) + 4 * h];
    }
}

hipError_t init_mlp_fused_kernels() {
    hipError_t e = hipSuccess;
    const int bias = (kMaxHidden + 32) * (int)sizeof(float);
#define DD_ATTR(DV)                                                                                          \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                MlpCfg<DV>::RING + bias);
    DD_ATTR(64) DD_ATTR(128) DD_ATTR(256) DD_ATTR(512)
#undef DD_ATTR
    return e;
}

hipError_t launch_mlp_fused(const MlpFusedArgs& a, int D, hipStream_t s) {
    switch (D) {
        case 64: return launch_d<64>(a, s);
        case 128: return launch_d<128>(a, s);
        case 256: return launch_d<256>(a, s);
        case 512: return launch_d<512>(a, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace dd
