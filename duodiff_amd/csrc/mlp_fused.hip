// Fused tail of a U-ViT block for gfx950, ONE launch, bf16 MFMA operands, fp32 accumulation and fp32 residual stream:
//     x1 = x + proj(ao) + b          (optional; reference models/uvit.py:166 + the residual add of :206)
//     h2 = norm2(x1)                 (optional, from the accumulators; :207)
//     y  = x1 + fc2( GELU_erf( fc1(h2) + b1 ) ) + b2        (:86-92, :207)
//     outputs: y (fp32, in place), bf16(y) (long-skip / next GEMM operand), norm1_next(y) in bf16 (optional; next block's :206)
// Neither x1, h2 nor the hidden activation [M, 4D] (135 MB per block at B = 128) ever exists in HBM or LDS.
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

constexpr int kMaxHidden = 4096;   // bias table in LDS next to the 128 KB ring
constexpr int kProjRowsLds = 4 * 32 * (1024 + 32);   // proj_rows_kernel: one padded 1 KB strip per LDS-DMA instruction
constexpr int kGroups = 16;        // hidden-split ways of an extra-token tile (32 measured slower: more slab traffic, same fixed costs)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_dst, 16, 0, 0);
}

// Same, addressed as (uniform base) + (32-bit lane offset): the base is made opaque so that hipcc does not hoist one 64-bit VGPR pointer per
// request out of the hot loop (16 pairs live across it otherwise -- the registers the loop does not have).  What it emits is still the VGPR-pair
// form `global_load_lds_dwordx4 v[n:n+1], off offset:imm` (one address computation per four requests through the immediate offset); the true
// scalar-base form `voff, s[base:base+1]` needs asm (rowlin.hip, gemm.hip, attention.hip use it) and measured neutral HERE (2 281 vs 2 246 cycles
// per chunk, 156.1 vs 156.3 us per launch, profiles/r05/ab_round5.txt): this loop hides its request issue already.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    asm volatile("" : "+s"(p));
    return p;
}
__device__ __forceinline__ void glds16u(const char* ubase, unsigned voff, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)(ubase + voff), (lptr_t)lds_dst, 16, 0, 0);
}
// Request j of a block (1 KB apart in the image and in the ring): the instruction's immediate offset -- it applies to the
// global AND the LDS address -- carries (j & 3) KB, so four requests share one 64-bit address computation and one M0
// setup.  At one wave per SIMD those were two of the three instructions every request cost (15 us of a fused launch).
template <int J>
__device__ __forceinline__ void glds16u_j(const char* ublock, unsigned voff, char* lds_block) {
    constexpr int HI = (J >> 2) * 4096, IMM = (J & 3) * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)(ublock + HI + voff), (lptr_t)(lds_block + HI), 16, IMM, 0);
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
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ void lds_frag(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_quad(f32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

// exact-erf GELU for a bf16-rounded result: gelu(v) = v * (0.5 + s * P'(s^2)), s = med3(v, +-3.8), P' = P / 2 with P the degree-6
// minimax polynomial of erf(s/sqrt2)/s (same coefficients as the GEMM epilogue in gemm.hip: |erf error| <= 1.3e-4, GELU
// abs error <= 2.4e-4).  TWO values are evaluated together as interleaved chains (no instruction reads the result of the
// one in front of it), cut into 8 pieces of at most 3 VALU instructions (21 in all: at one wave per SIMD the loop is bound by
// vector issue -- a lone wave issues a VALU instruction every ~8 clocks, profiles/r02/exp_rate.txt -- so every instruction
// counts); one piece sits in the gap behind one MFMA.
struct GeluConst { float hi, c5; };   // 3.8 and the s^12 coefficient live in VGPRs (VOP3 / fmamk take no second literal)
struct GeluPair { float sa, s2a, pa, ha, sb, s2b, pb, hb; };

// ONE gap of the instruction stream as ONE asm statement:
//     s_waitcnt lgkmcnt(LG) ; MFMA ; [ds_read_b128 of the fragment PD steps ahead, into the register the MFMA just read]
//     ; [piece K of the GELU pair (va, vb)]
// KIND 0: S^T += W1 fragment . X^T, accumulator in VGPRs;  KIND 1: Y^T tile += W2 fragment . P^T, accumulator in AGPRs.
// Every operand is declared for every variant (unused ones cost nothing); the GELU registers are read-write throughout.
#define DD_S_MFMA_W "s_waitcnt lgkmcnt(%[lg])\n\tv_mfma_f32_32x32x16_bf16 %[acc], %[wa], %[xb], %[acc]"
#define DD_S_MFMA_N "v_mfma_f32_32x32x16_bf16 %[acc], %[wa], %[xb], %[acc]"     // LG < 0: the gap in front already waited for this fragment
#define DD_S_READ "\n\tds_read_b128 %[wa], %[la] offset:%[lo]"
#define DD_S_G0 "\n\tv_med3_f32 %[sa], %[va], %[kn], %[kh]\n\tv_med3_f32 %[sb], %[vb], %[kn], %[kh]\n\tv_mul_f32 %[s2a], %[sa], %[sa]"
#define DD_S_G1 "\n\tv_mul_f32 %[s2b], %[sb], %[sb]\n\tv_fmamk_f32 %[pa], %[s2a], 0x331d7172, %[kc]\n\tv_fmamk_f32 %[pb], %[s2b], 0x331d7172, %[kc]"
#define DD_S_G2 "\n\tv_fmaak_f32 %[pa], %[pa], %[s2a], 0x387e87ac\n\tv_fmaak_f32 %[pb], %[pb], %[s2b], 0x387e87ac\n\tv_fmaak_f32 %[pa], %[pa], %[s2a], 0xba743309"
#define DD_S_G3 "\n\tv_fmaak_f32 %[pb], %[pb], %[s2b], 0xba743309\n\tv_fmaak_f32 %[pa], %[pa], %[s2a], 0x3c18a4c9\n\tv_fmaak_f32 %[pb], %[pb], %[s2b], 0x3c18a4c9"
#define DD_S_G4 "\n\tv_fmaak_f32 %[pa], %[pa], %[s2a], 0xbd869818\n\tv_fmaak_f32 %[pb], %[pb], %[s2b], 0xbd869818\n\tv_fmaak_f32 %[pa], %[pa], %[s2a], 0x3ecc1f5c"
#define DD_S_G5 "\n\tv_fmaak_f32 %[pb], %[pb], %[s2b], 0x3ecc1f5c\n\tv_fmaak_f32 %[ha], %[sa], %[pa], 0x3f000000\n\tv_fmaak_f32 %[hb], %[sb], %[pb], 0x3f000000"
#define DD_S_G6 "\n\tv_mul_f32 %[ha], %[ha], %[va]\n\tv_mul_f32 %[hb], %[hb], %[vb]"
#define DD_S_G7 "\n\tv_cvt_pk_bf16_f32 %[out], %[ha], %[hb]"
#define DD_GAP_ASM(STR, ACC_C)                                                                                          \
    asm volatile(STR                                                                                                    \
                 : [acc] ACC_C(acc), [wa] "+v"(wa), [sa] "+v"(r.sa), [s2a] "+v"(r.s2a), [pa] "+v"(r.pa), [ha] "+v"(r.ha),  \
                   [sb] "+v"(r.sb), [s2b] "+v"(r.s2b), [pb] "+v"(r.pb), [hb] "+v"(r.hb), [out] "+v"(out)                 \
                 : [xb] "v"(xb), [la] "v"(la), [va] "v"(va), [vb] "v"(vb), [kn] "s"(-3.8f), [kh] "v"(k.hi), [kc] "v"(k.c5), \
                   [lg] "i"(LG), [lo] "i"(LO))
#define DD_GAP_SEL(GSTR, ACC_C)                                                    \
    if constexpr (LG >= 0) {                                                      \
        if constexpr (READ) DD_GAP_ASM(DD_S_MFMA_W DD_S_READ GSTR, ACC_C);        \
        else DD_GAP_ASM(DD_S_MFMA_W GSTR, ACC_C);                                 \
    } else {                                                                      \
        if constexpr (READ) DD_GAP_ASM(DD_S_MFMA_N DD_S_READ GSTR, ACC_C);        \
        else DD_GAP_ASM(DD_S_MFMA_N GSTR, ACC_C);                                 \
    }
#define DD_GAP_K(KK, GSTR, ACC_C) else if constexpr (K == KK) { DD_GAP_SEL(GSTR, ACC_C) }
template <int KIND, int LG, bool READ, int LO, int K>
__device__ __forceinline__ void gap_stmt(f32x16& acc, bf16x8& wa, const bf16x8& xb, unsigned la, float va, float vb,
                                         const GeluConst& k, GeluPair& r, unsigned& out) {
    if constexpr (KIND == 0) {
        if constexpr (K < 0) { DD_GAP_SEL("", "+v") }
        DD_GAP_K(0, DD_S_G0, "+v") DD_GAP_K(1, DD_S_G1, "+v") DD_GAP_K(2, DD_S_G2, "+v") DD_GAP_K(3, DD_S_G3, "+v")
        DD_GAP_K(4, DD_S_G4, "+v") DD_GAP_K(5, DD_S_G5, "+v") DD_GAP_K(6, DD_S_G6, "+v") DD_GAP_K(7, DD_S_G7, "+v")
    } else {
        if constexpr (K < 0) { DD_GAP_SEL("", "+a") }
        DD_GAP_K(0, DD_S_G0, "+a") DD_GAP_K(1, DD_S_G1, "+a") DD_GAP_K(2, DD_S_G2, "+a") DD_GAP_K(3, DD_S_G3, "+a")
        DD_GAP_K(4, DD_S_G4, "+a") DD_GAP_K(5, DD_S_G5, "+a") DD_GAP_K(6, DD_S_G6, "+a") DD_GAP_K(7, DD_S_G7, "+a")
    }
}

// A gap without a GELU piece, accumulator in AGPRs, with only the operands it uses (the SKIP phases: 2 NT F statements; the
// generic statement above declares the GELU constants as operands -- an SGPR float among them -- whether it uses them or not)
template <int LG, bool READ, int LO>
__device__ __forceinline__ void gap_plain(f32x16& acc, bf16x8& wa, const bf16x8& xb, unsigned la) {
    if constexpr (LG >= 0) {
        if constexpr (READ) asm volatile(DD_S_MFMA_W DD_S_READ : [acc] "+a"(acc), [wa] "+v"(wa) : [xb] "v"(xb), [la] "v"(la), [lg] "i"(LG), [lo] "i"(LO));
        else asm volatile(DD_S_MFMA_W : [acc] "+a"(acc), [wa] "+v"(wa) : [xb] "v"(xb), [lg] "i"(LG));
    } else {
        if constexpr (READ) asm volatile(DD_S_MFMA_N DD_S_READ : [acc] "+a"(acc), [wa] "+v"(wa) : [xb] "v"(xb), [la] "v"(la), [lo] "i"(LO));
        else asm volatile(DD_S_MFMA_N : [acc] "+a"(acc), [wa] "+v"(wa) : [xb] "v"(xb));
    }
}

// the same with the accumulator in VGPRs (the QKV phases: the tile is converted and stored by VALU code)
template <int LG, bool READ, int LO>
__device__ __forceinline__ void gap_plain_v(f32x16& acc, bf16x8& wa, const bf16x8& xb, unsigned la) {
    if constexpr (LG >= 0) {
        if constexpr (READ) asm volatile(DD_S_MFMA_W DD_S_READ : [acc] "+v"(acc), [wa] "+v"(wa) : [xb] "v"(xb), [la] "v"(la), [lg] "i"(LG), [lo] "i"(LO));
        else asm volatile(DD_S_MFMA_W : [acc] "+v"(acc), [wa] "+v"(wa) : [xb] "v"(xb), [lg] "i"(LG));
    } else {
        if constexpr (READ) asm volatile(DD_S_MFMA_N DD_S_READ : [acc] "+v"(acc), [wa] "+v"(wa) : [xb] "v"(xb), [la] "v"(la), [lo] "i"(LO));
        else asm volatile(DD_S_MFMA_N : [acc] "+v"(acc), [wa] "+v"(wa) : [xb] "v"(xb));
    }
}

// the same GELU pieces as statements of their own (D < 512: several pieces per gap)
template <int K>
__device__ __forceinline__ void gelu_piece(float va, float vb, const GeluConst& k, GeluPair& r, unsigned& out) {
    if constexpr (K == 0) asm volatile(DD_S_G0 : [sa] "+v"(r.sa), [sb] "+v"(r.sb), [s2a] "+v"(r.s2a) : [va] "v"(va), [vb] "v"(vb), [kn] "s"(-3.8f), [kh] "v"(k.hi));
    else if constexpr (K == 1) asm volatile(DD_S_G1 : [s2b] "+v"(r.s2b), [pa] "+v"(r.pa), [pb] "+v"(r.pb) : [sb] "v"(r.sb), [s2a] "v"(r.s2a), [kc] "v"(k.c5));
    else if constexpr (K == 2) asm volatile(DD_S_G2 : [pa] "+v"(r.pa), [pb] "+v"(r.pb) : [s2a] "v"(r.s2a), [s2b] "v"(r.s2b));
    else if constexpr (K == 3) asm volatile(DD_S_G3 : [pa] "+v"(r.pa), [pb] "+v"(r.pb) : [s2a] "v"(r.s2a), [s2b] "v"(r.s2b));
    else if constexpr (K == 4) asm volatile(DD_S_G4 : [pa] "+v"(r.pa), [pb] "+v"(r.pb) : [s2a] "v"(r.s2a), [s2b] "v"(r.s2b));
    else if constexpr (K == 5) asm volatile(DD_S_G5 : [pb] "+v"(r.pb), [ha] "+v"(r.ha), [hb] "+v"(r.hb) : [s2b] "v"(r.s2b), [sa] "v"(r.sa), [sb] "v"(r.sb), [pa] "v"(r.pa));
    else if constexpr (K == 6) asm volatile(DD_S_G6 : [ha] "+v"(r.ha), [hb] "+v"(r.hb) : [va] "v"(va), [vb] "v"(vb));
    else asm volatile(DD_S_G7 : [out] "+v"(out) : [ha] "v"(r.ha), [hb] "v"(r.hb));
}

// 16-byte global load straight into accumulator registers (AGPRs are legal vector-memory destinations on gfx950): the 64
// loads of a lane's row are all in flight at once without holding a single VGPR.  hipcc does not count asm loads: the
// caller waits (vmcnt) and fences the destinations before using them.
template <int OFF>
__device__ __forceinline__ void load_quad_agpr(f32x4& q, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(q) : "v"(p), "i"(OFF));
}

// Pin an accumulator tile to its AGPRs at this point of the program: hipcc treats the tile as rewritten, so VGPR copies
// made for the LayerNorm / epilogue arithmetic die here instead of piling up (256 live values would spill to scratch).
__device__ __forceinline__ void acc_pin(f32x16& y) { asm volatile("" : "+a"(y)); }

// ... all tiles of a wave in ONE statement (asm operand lists take no pack expansion)
template <int NT>
__device__ __forceinline__ void pin_tiles(f32x16 (&y)[NT]) {
    static_assert(NT == 2 || NT == 4 || NT == 8 || NT == 16, "D in {64, 128, 256, 512}");
    if constexpr (NT == 2) asm volatile("" : "+a"(y[0]), "+a"(y[1]));
    else if constexpr (NT == 4) asm volatile("" : "+a"(y[0]), "+a"(y[1]), "+a"(y[2]), "+a"(y[3]));
    else if constexpr (NT == 8)
        asm volatile("" : "+a"(y[0]), "+a"(y[1]), "+a"(y[2]), "+a"(y[3]), "+a"(y[4]), "+a"(y[5]), "+a"(y[6]), "+a"(y[7]));
    else
        asm volatile("" : "+a"(y[0]), "+a"(y[1]), "+a"(y[2]), "+a"(y[3]), "+a"(y[4]), "+a"(y[5]), "+a"(y[6]), "+a"(y[7]), "+a"(y[8]),
                     "+a"(y[9]), "+a"(y[10]), "+a"(y[11]), "+a"(y[12]), "+a"(y[13]), "+a"(y[14]), "+a"(y[15]));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// LayerNorm statistics of a row that is spread over the two lane halves (lane, lane ^ 32), D / 2 columns each, from ONE
// pass over the registers -- without the cancellation of E[x^2] - mean^2: every lane accumulates s = sum(x - c) and
// q = sum((x - c)^2) around a shift c that is one of its OWN elements (so (mean - c)^2 <= n var and the subtraction
// q - s^2 / n loses at most ~n eps of var, whatever offset the row carries), and the two halves are combined exactly
// (Chan et al.).  Both lanes of a row compute bit-identical results.  rstd = 1 / sqrt(var + 1e-5), biased variance
// (torch.nn.LayerNorm, reference models/uvit.py:185-189).
template <int D>
__device__ __forceinline__ void ln_stats_shifted(float c, float s, float q, float& mean, float& rstd) {
    constexpr float n = (float)(D / 2);
    const float mh = c + s / n;
    const float m2h = q - s * s / n;
    // v_permlane32_swap of a value with itself: afterwards element 0 is the LOWER half's value and element 1 the UPPER
    // half's, on every lane of the row's pair -- no lane address (a ds_bpermute address computed here would stay live, or
    // be spilled, across the whole chunk loop for the epilogue's use), and both lanes evaluate the same expression
    const auto pm = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mh), __builtin_bit_cast(unsigned, mh), false, false);
    const auto p2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m2h), __builtin_bit_cast(unsigned, m2h), false, false);
    // (elements copied to scalars first: __builtin_bit_cast applied directly to a vector-element expression pm[1] reads
    // element 0 with this clang -- the difference below came out as v - v)
    const unsigned u_lo = pm[0], u_hi = pm[1], w_lo = p2[0], w_hi = p2[1];
    const float m_lo = __builtin_bit_cast(float, u_lo), m_hi = __builtin_bit_cast(float, u_hi);
    const float q_lo = __builtin_bit_cast(float, w_lo), q_hi = __builtin_bit_cast(float, w_hi);
    mean = 0.5f * (m_lo + m_hi);
    const float d = m_lo - m_hi;
    const float m2 = (q_lo + q_hi) + d * d * (0.5f * n);
    const float var = m2 / (float)D;
    rstd = 1.0f / sqrtf((var > 0.f ? var : 0.f) + 1e-5f);
}

template <int D>
struct MlpCfg {
    static constexpr int NT = D / 32;          // 32-column output tiles of one wave
    static constexpr int KS = D / 16;          // k-steps (16 wide) of the first GEMM
    static constexpr int F = D / 16;           // 1 KB fragments per ring block (W1 chunk: KS, W2 chunk: 2 * NT)
    static constexpr int BLK = F * 1024;       // bytes per ring block
    static constexpr int RING = 4 * BLK;
    static constexpr int FPW = F / 4;          // fragments each of the 4 waves DMAs per block
};

// PROJ (main tiles, LNIN mode): the attention output projection of the block runs in front of the MLP inside the same
// launch -- x1 = x + ao . Wproj^T + bproj (reference models/uvit.py:206, Attention.proj :166) accumulates on top of x in the
// output accumulators, norm2 is taken from those registers, the MLP accumulates on top again: x1 never exists in HBM.
// SKIP (main tiles of a PROJ launch): the NEXT block's skip_linear + norm1 run behind the MLP inside the same launch --
// x' = [y | skip] . Wskip^T + bskip (reference models/uvit.py:196-200: the block output y only feeds this Linear, it is never
// stored), norm1(x') as bf16 (:206).  y goes from the accumulators straight into MFMA B fragments (as norm2's output does
// in front of fc1), the accumulators restart from bskip, and 2 NT more phases stream the skip weight blocks that follow the
// MLP blocks in the image: NT blocks of the y half (one column tile x all its k-steps each), then the long-skip operand's
// half in two passes of NT/2 blocks (two column tiles x half the k-steps each) -- the operand's rows are fetched half at
// a time into registers that are free at that point, each half a pass ahead of its use.
// QKV (main tiles of a PROJ launch): the NEXT block's attn.qkv Linear (no bias) runs behind everything else -- its input
// norm1(x) never leaves the registers.  Order at the end of the launch: LayerNorm statistics from the accumulators, norm1
// straight into MFMA B fragments, 3D/32 phases (one 32-column tile of q | k | v each, weights streamed through the ring like
// every other block: the qkv blocks close the image) into two alternating VGPR accumulators -- tile t-1 is converted to bf16
// and stored (head-major, 64 contiguous bytes per row and phase) in the second half of phase t, so 100 MB of qkv leave the
// chip under MFMA work -- and only then the fp32 rows x (and their bf16 copy) are stored.
template <int D, bool LNIN, bool PARTIAL, bool PROJ, bool SKIP = false, bool QKV = false, bool TAP = false>
__device__ __forceinline__ void mlp_body(const MlpFusedArgs& a, char* smem, const int tile_idx, const int c0, int c1, const int slab) {
    using C = MlpCfg<D>;
    static_assert(!PROJ || (LNIN && C::NT % 4 == 0), "proj fusion: the LayerNorm-in kernel, D % 128 == 0");
    static_assert(!SKIP || (PROJ && !PARTIAL), "skip fusion rides on the proj-fused main tiles");
    static_assert(!QKV || (PROJ && !PARTIAL), "qkv fusion rides on the proj-fused main tiles");
    float* b1s = reinterpret_cast<float*>(smem + C::RING);           // [hidden + 32], in accumulator-register order per chunk
    float* vecs = b1s + (a.nchunks + 1) * 32;                        // 7 x [D]: ln_in gamma, beta | ln_out gamma, beta | b2 | bproj | bskip
    // weight stream: [nproj blocks of Wproj][W1(0) W2(0) W1(1) W2(1) ...]; stream position p lives in ring slot p & 3
    // (nproj % 4 == 0), so the MLP part keeps "block b in slot b & 3" with or without the projection in front
    const char* const wproj = a.wimg;
    const char* const wmlp = a.wimg + (size_t)a.nproj * C::BLK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;

    // logical row of this lane -> physical token row (clamped to a valid row; stores are masked by row_ok).  Evaluated
    // once for the prologue and again for the epilogue from an opaque copy of the lane id: kept live across the chunk
    // loop these values (and the per-lane pointers derived from them) are what tips the loop over 256 VGPRs into spills.
    auto row_of = [&](bool& ok) -> long long {
        unsigned l = (unsigned)threadIdx.x;
        asm volatile("" : "+v"(l));
        const int local = (int)l / 64 * 32 + ((int)l & 31);
        const int idx = tile_idx * (PARTIAL ? a.prows : 128) + local;      // (hidden-split tiles hold prows <= 128 rows: waves beyond them idle along)
        if constexpr (!PARTIAL) {
            ok = idx < a.n_main;
            const int p = ok ? idx : 0, b = p / a.tok_n;
            return (long long)b * a.tok_l + a.tok_e + (p - b * a.tok_n);
        } else {
            ok = local < a.prows && idx < a.n_extra;
            const int q = ok ? idx : 0, b = q / a.tok_e;
            return (long long)b * a.tok_l + (q - b * a.tok_e);
        }
    };
    auto half_of = [&]() -> int {     // lane >> 5, opaque (same reason)
        unsigned l = (unsigned)threadIdx.x;
        asm volatile("" : "+v"(l));
        return (int)(l >> 5) & 1;
    };

    const unsigned dma_voff = (wave * C::FPW) * 1024 + lane * 16;    // this lane's 16 bytes of request j: block + j * 1024 + dma_voff
    // ---- LDS-DMA of one ring block.  Stream position b: chunk b>>1, W1 block if b is even, W2 block if odd.
    auto dma_block = [&](int b, int slot) {
        const char* src = uniform_ptr((PROJ ? wproj : wmlp) + (size_t)b * C::BLK);
        char* dst = smem + slot * C::BLK + (wave * C::FPW) * 1024;
        [&]<int... J>(std::integer_sequence<int, J...>) { (glds16u_j<J>(src, dma_voff, dst), ...); }(std::make_integer_sequence<int, C::FPW>{});
    };

    // prologue: the first three blocks of the stream (W1(c0), W2(c0), W1(c0+1); with PROJ the first three Wproj blocks) in
    // flight while the X fragments and the bias table are fetched.  c0 is even (mlp_fused_plan), so block b always lives
    // in ring slot b & 3: W1(c) in slot 0 / 2, W2(c) in slot 1 / 3.
    const int pb0 = PROJ ? 0 : 2 * c0;     // (PROJ: the projection's blocks lead the stream whatever hidden range follows)
    dma_block(pb0, 0);
    dma_block(pb0 + 1, 1);
    dma_block(pb0 + 2, 2);
    if constexpr (!PROJ) {   // slot 3 stands in for "W2 of chunk c0-1": zeros, so that the first iteration's GEMM2 adds nothing
        // (PROJ: the last Wproj block sits there -- finite, and multiplied by the all-zero activations of "chunk -1")
        f32x4* z = reinterpret_cast<f32x4*>(smem + 3 * C::BLK);
        for (int i = tid; i < C::BLK / 16; i += 256) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    {   // bias table and per-column vectors -> LDS (the LayerNorm / epilogue arithmetic reads the vectors as broadcast
        // ds_read_b128 -- every lane of a half wants the same 16 bytes -- not as 300+ vector-memory instructions per lane).
        // Every global load is issued before the first LDS write: one memory round trip, not one per vector (hipcc keeps
        // separate load -> ds_write loops serial, which cost each workgroup ~8 round trips before its first MFMA).
        constexpr int TB = kMaxHidden / 4 / 256, VQ = 7 * (D / 4), VI = (VQ + 255) / 256;
        const int nb1 = a.nchunks * 8;
        f32x4 tb[TB], tv[VI];
        bool tvok[VI];
#pragma unroll
        for (int k = 0; k < TB; ++k) {
            const int i = tid + 256 * k;
            tb[k] = reinterpret_cast<const f32x4*>(a.b1p)[i < nb1 ? i : 0];
        }
#pragma unroll
        for (int k = 0; k < VI; ++k) {
            const int item = tid + 256 * k, v = item / (D / 4), i = item - v * (D / 4);
            const float* sp = v == 0 ? a.ln_in_g : v == 1 ? a.ln_in_b : v == 2 ? a.ln_out_g : v == 3 ? a.ln_out_b : v == 4 ? a.b2 : v == 5 ? a.bproj : a.bskip;
            tvok[k] = item < VQ && sp != nullptr;
            tv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (tvok[k]) tv[k] = reinterpret_cast<const f32x4*>(sp)[i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < TB; ++k)
            if (tid + 256 * k < nb1) reinterpret_cast<f32x4*>(b1s)[tid + 256 * k] = tb[k];
#pragma unroll
        for (int k = 0; k < VI; ++k)
            if (tvok[k]) reinterpret_cast<f32x4*>(vecs)[tid + 256 * k] = tv[k];
    }
    const float* lg_in = vecs + 4 * h;                 // this lane's column offset inside a quad pair
    const float* lb_in = vecs + D + 4 * h;
    bool row_ok_pro;
    const long long row_pro = row_of(row_ok_pro);

    bf16x8 xf[C::KS];
    f32x16 Y[C::NT];
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (!LNIN) {
        const bf16_t* xr = a.X + row_pro * a.ldx + 8 * h;
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
#pragma unroll
        for (int t = 0; t < C::NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) Y[t][e] = 0.f;
        __syncthreads();   // (vmcnt(0): the prologue blocks have landed; bias table, vectors and zero block are visible)
    } else {
        // x rows in accumulator layout: register quad g of tile t = columns 32t + 8g + 4h .. +3 of the lane's row.
        // Main tiles keep x in the accumulators (the MLP output is added on top: no second read of x); hidden-split
        // tiles zero them after the normalisation (x is added once, by the reduce kernel).
        // This code runs at one wave per SIMD with nothing to hide latency behind: its length is its cost.
        const float* xr = a.xres + row_pro * D + 4 * h;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
        float cshift = 0.f;      // shift of the one-pass statistics: the lane's first element (ln_stats_shifted)
        // LA tiles of loads in flight, no more: sched_barrier keeps hipcc from hoisting all 64 loads (256 registers)
        // above the arithmetic (that version spilled); statistics in one pass on register quads (packed fp32 math)
        constexpr int LA = C::NT < 4 ? C::NT : (PARTIAL && !PROJ && C::NT >= 8) ? 8 : 4;   // hidden-split tiles without the projection: nothing else is live yet, and the workgroup is pure latency
        f32x4 xq[LA][4];
        const float* lbp = vecs + 5 * D + 4 * h;     // (PROJ) attn.proj bias
        if constexpr (PROJ) {
            // attention output rows of this wave as B fragments (natural k order; they live in the registers the
            // normalised rows take over afterwards)
            const bf16_t* ar = a.ao + row_pro * D + 8 * h;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(ar + 16 * ks);
            __syncthreads();                          // the bias vector is in LDS (and the first Wproj blocks have landed)
        }
#pragma unroll
        for (int t = 0; t < LA - 1; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) xq[t][g] = *reinterpret_cast<const f32x4*>(xr + 32 * t + 8 * g);
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            if (t + LA - 1 < C::NT) {
#pragma unroll
                for (int g = 0; g < 4; ++g) xq[(t + LA - 1) % LA][g] = *reinterpret_cast<const f32x4*>(xr + 32 * (t + LA - 1) + 8 * g);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 q = xq[t % LA][g];
                if constexpr (PROJ) q += *reinterpret_cast<const f32x4*>(lbp + 32 * t + 8 * g);   // x + bproj: the projection accumulates on top
                if constexpr (!PROJ) {                     // (PROJ: the statistics are taken from x1 after the projection)
                    if (t == 0 && g == 0) cshift = q[0];
                    const f32x4 dq = q - cshift;
                    s4 += dq;
                    q4 += dq * dq;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) Y[t][4 * g + e] = q[e];
            }
            acc_pin(Y[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PROJ) {
            // ---- x1 = x + bproj + ao . Wproj^T : NT phases of F MFMAs, phase t = output columns 32t .. 32t+31 = block t of
            // the stream, fed by one fragment queue PDp deep that runs on across the phase boundaries (as in the chunk loop
            // below).  In the middle of phase t: block t+1 landed (in-order groups, one younger one may be outstanding) |
            // barrier | request block t+3 into the slot block t-1 has left.  The queue first touches block t+1 after that
            // point (PDp <= F/2).  The last three requests are the MLP's first blocks.
            const unsigned lds_lo_p = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
            const unsigned lds_hi_p = lds_lo_p + 65536u;
            constexpr int PDp = C::F / 2 < 8 ? C::F / 2 : 8;
            constexpr int NGp = C::NT * C::F;
            bf16x8 wp[PDp];
            GeluPair gr0{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const GeluConst gk0{0.f, 0.f};
            unsigned none = 0;
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (lds_frag<J * 1024>(wp[J], lds_lo_p), ...);
            }(std::make_integer_sequence<int, PDp>{});
            [&]<int... GI>(std::integer_sequence<int, GI...>) {
                ([&] {
                    constexpr int g = GI, t = g / C::F, f = g % C::F;
                    if constexpr (f == C::F / 2) {
                        wait_vmcnt<C::FPW>();
                        __builtin_amdgcn_s_barrier();
                        // stream position t + 3: Wproj block, or (t + 3 >= NT) block 2 c0 + t + 3 - NT of the MLP part (c0 = 0 for main tiles;
                        // a hidden-split workgroup continues with ITS hidden range: c0 is even, so the ring slot parity holds)
                        const char* src = uniform_ptr(a.wimg + (size_t)(t + 3) * C::BLK + (t + 3 >= C::NT ? (size_t)(2 * c0) * C::BLK : (size_t)0));
                        char* dst = smem + ((t + 3) & 3) * C::BLK + (wave * C::FPW) * 1024;
                        [&]<int... J>(std::integer_sequence<int, J...>) { (glds16u_j<J>(src, dma_voff, dst), ...); }(std::make_integer_sequence<int, C::FPW>{});
                    }
                    // one counted wait per pair of gaps, as in the chunk loop: the even gap waits for the odd gap's fragment too
                    constexpr int gn = g + PDp, left = NGp - 1 - g;
                    constexpr int lg_self = left < PDp - 1 ? left : PDp - 1, lg_next = left - 1 < PDp - 1 ? left - 1 : PDp - 1;
                    constexpr int LG = (g & 1) ? -1 : (left >= 1 ? lg_next - (left >= PDp ? 1 : 0) : lg_self);   // (this gap's own read is not issued yet)
                    constexpr bool RD = gn < NGp;
                    constexpr int LO = RD ? ((gn / C::F) & 3) * C::BLK + (gn % C::F) * 1024 : 0;
                    constexpr int LOA = LO < 65536 ? LO : LO - 65536;
                    gap_stmt<1, LG, RD, LOA, -1>(Y[t], wp[g % PDp], xf[f], LO < 65536 ? lds_lo_p : lds_hi_p, 0.f, 0.f, gk0, gr0, none);
                }(), ...);
            }(std::make_integer_sequence<int, NGp>{});
            mfma_drain();
            // statistics of x1 from the accumulators (one tuple copy per tile)
            s4 = f32x4{0.f, 0.f, 0.f, 0.f};
            q4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                acc_pin(Y[t]);
                const f32x16 yt = Y[t];
                if (t == 0) cshift = yt[0];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 q = {yt[4 * g], yt[4 * g + 1], yt[4 * g + 2], yt[4 * g + 3]};
                    const f32x4 dq = q - cshift;
                    s4 += dq;
                    q4 += dq * dq;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float mean, rstd;
        ln_stats_shifted<D>(cshift, (s4[0] + s4[1]) + (s4[2] + s4[3]), (q4[0] + q4[1]) + (q4[2] + q4[3]), mean, rstd);
        const float shift = -mean * rstd;
        __syncthreads();   // (vmcnt(0): the MLP's first three blocks have landed; bias table, vectors [and zero block] are visible)
        // k-step ks of fc1 = registers 8 (ks & 1) .. + 7 of tile ks >> 1: element j is column
        // 16 ks + 8 (j >> 2) + 4 h + (j & 3) -- the permuted k order the W1 image is packed in (mlp_fused_pack, kperm)
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            acc_pin(Y[t]);                           // (PROJ: the copies made for the statistics die here, not in scratch)
            const f32x16 yt = Y[t];                  // ONE copy of the tile out of the AGPRs (element-wise access re-reads all 16)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                const int ks = 2 * t + kq;
                unsigned u[4];
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) {
                    const int col = 16 * ks + 8 * gq, g = 2 * kq + gq;
                    const f32x4 q = {yt[4 * g], yt[4 * g + 1], yt[4 * g + 2], yt[4 * g + 3]};
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(lg_in + col), bv = *reinterpret_cast<const f32x4*>(lb_in + col);
                    const f32x4 v = (q * rstd + shift) * gv + bv;
                    u[2 * gq] = pack2(v[0], v[1]);
                    u[2 * gq + 1] = pack2(v[2], v[3]);
                }
                xf[ks] = __builtin_bit_cast(bf16x8, u32x4{u[0], u[1], u[2], u[3]});
            }
            if constexpr (PARTIAL && !PROJ) {        // hidden-split tiles accumulate a partial sum: the row only passed through
#pragma unroll
                for (int e = 0; e < 16; ++e) Y[t][e] = 0.f;
                acc_pin(Y[t]);
            } else if constexpr (PARTIAL) {
                // ... with the projection in front, every group of a tile has computed x1 = x + proj(ao) + b for its rows (the LayerNorm needs all
                // of it); the FIRST group's slab carries it, so that the reduce launch sets x = b2 + sum of the slabs (MlpFusedArgs::reduce_set)
                // and no launch writes x1 while another group may still read x.  An exact 1 / 0 factor, not a branch around 256 accumulators.
                const float keep1 = c0 == 0 ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) Y[t][e] = yt[e] * keep1;
                acc_pin(Y[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // LDS addressing: one per-lane base (+ a second one 64 KB up: ds offsets are 16 bits), compile-time offsets
    const unsigned lds_lo = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
    const unsigned lds_hi = lds_lo + 65536u;
    const unsigned bias_lo = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + C::RING + h * 64;
    constexpr int PD = C::F < 8 ? C::F : 8;    // fragment reads in flight ahead of their MFMA: covers ~250 cycles of LDS latency
    const GeluConst gk{3.8f, 0.5f * -4.544908101e-06f};     // (the polynomial's coefficients are halved: 0.5 * erf(s / sqrt2) / s)

    // LDS offset of fragment f of ring slot `slot` relative to lds_lo (compile-time)
    auto frag_off = [](int slot, int f) constexpr { return slot * C::BLK + f * 1024; };
    auto frag = [&](auto off_tag, bf16x8& q) {
        constexpr int OFF = decltype(off_tag)::value;
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

    GeluPair gr{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned pw_none = 0;
    f32x16 sA, sB;
    bf16x8 wq[PD];
    bias_init(c0, sA);
    // S of the first chunk from ring slot 0, with a fragment queue of its own
    [&]<int... J>(std::integer_sequence<int, J...>) { (frag(std::integral_constant<int, frag_off(0, J)>{}, wq[J]), ...); }
    (std::make_integer_sequence<int, PD>{});
    [&]<int... FI>(std::integer_sequence<int, FI...>) {
        ([&] {
            constexpr int f = FI, left = C::F - 1 - f, LG = left < PD - 1 ? left : PD - 1;
            constexpr bool RD = f + PD < C::F;
            constexpr int LO = RD ? frag_off(0, f + PD) : 0;
            gap_stmt<0, LG, RD, LO, -1>(sA, wq[f % PD], xf[f], lds_lo, 0.f, 0.f, gk, gr, pw_none);
        }(), ...);
    }(std::make_integer_sequence<int, C::F>{});
    // the continuous fragment queue of the loop starts here: first PD fragments of "W2 of chunk c0 - 1" (slot 3: zeros)
    [&]<int... J>(std::integer_sequence<int, J...>) { (frag(std::integral_constant<int, frag_off(3, J)>{}, wq[J]), ...); }
    (std::make_integer_sequence<int, PD>{});

    // One chunk c, PAR = c & 1.  s_cur = S of chunk c (complete), s_next receives S of chunk c+1, p_prev = GELU of chunk
    // c-1 (zeros for the first chunk), p_out receives the GELU of chunk c.  No branch inside: the last chunk computes a
    // throw-away S from the padded image.
    //
    // Instruction stream: 2F gaps (gap_stmt) -- F of GEMM2 for chunk c-1 (block W2(c-1)), then F of GEMM1 for chunk c+1
    // (block W1(c+1)) -- fed by ONE fragment queue PD deep that never drains: it runs on across the phase boundary and
    // into the next iteration's first block W2(c).  Behind each MFMA: the read PD fragments ahead and one GELU piece of
    // chunk c.  Two raw barriers per iteration order the ring; neither sits in front of a fragment read:
    //   E (before gap 0):  wait "W1(c+1) landed" | barrier | then request W1(c+2) into the slot W1(c) has left
    //   M (before gap F):  wait "W2(c)   landed" | barrier | then request W2(c+1) into the slot W2(c-1) has left
    // i.e. every block is confirmed half an iteration before its first fragment read and requested a full iteration
    // before that.  DMA groups (FPW instructions per wave) retire in issue order and at most two are outstanding at a
    // wait, so vmcnt(FPW) = "the older one has landed"; the barrier extends that to every wave's pieces and orders the
    // slot hand-over.  Nothing else in the loop touches vmcnt.  The FPW requests of a block are spread over the phase.
    constexpr int PPG = 32 / C::F;             // GELU pieces per gap: 64 pieces (8 pairs x 8) over 2F gaps
    constexpr int KB = C::F / 2;               // the 4 bias reads of s_next are issued behind gap KB
    constexpr int DSTEP = C::F / C::FPW;       // one LDS-DMA request every DSTEP gaps (4)
    auto iteration = [&](auto par_tag, int c, f32x16& s_cur, f32x16& s_next, const bf16x8 (&p_prev)[2], bf16x8 (&p_out)[2]) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr int SLOT_A = PAR ? 1 : 3;        // W2 of chunk c-1: block 2c-1
        constexpr int SLOT_B = PAR ? 0 : 2;        // W1 of chunk c+1: block 2c+2
        constexpr int SLOT_N = PAR ? 3 : 1;        // W2 of chunk c: block 2c+1, the next iteration's first block
        constexpr int NG = 2 * C::F;
        unsigned pw[8];
        const char* src_e = uniform_ptr(wmlp + (size_t)(2 * c + 4) * C::BLK);   // W1(c+2) -> slot of W1(c)
        const char* src_m = uniform_ptr(wmlp + (size_t)(2 * c + 3) * C::BLK);   // W2(c+1) -> slot of W2(c-1)
        char* dst_e = smem + (PAR ? 2 : 0) * C::BLK + (wave * C::FPW) * 1024;
        char* dst_m = smem + (PAR ? 1 : 3) * C::BLK + (wave * C::FPW) * 1024;
        [&]<int... GI>(std::integer_sequence<int, GI...>) {
            ([&] {
                constexpr int g = GI;
                if constexpr (g == 0 || g == C::F) {
                    wait_vmcnt<C::FPW>();
                    __builtin_amdgcn_s_barrier();
                }
                // fragment read PD gaps ahead: this iteration's slots, or the next iteration's first block
                constexpr int gn = g + PD;
                constexpr int LO = gn < C::F ? frag_off(SLOT_A, gn) : gn < NG ? frag_off(SLOT_B, gn - C::F) : frag_off(SLOT_N, gn - NG);
                constexpr int LOA = LO < 65536 ? LO : LO - 65536;
                const unsigned la = LO < 65536 ? lds_lo : lds_hi;
                // One counted wait per PAIR of gaps (at one wave per SIMD every instruction costs an issue slot the MFMA pipe
                // waits for): the even gap waits until the fragment of the odd gap behind it has landed (LDS reads return
                // in order, so its own has too).  Younger than fragment g+1 at that point: PD - 2 fragment reads, plus the 4
                // bias reads issued behind gap KB for the fragments requested before them.
                constexpr int lg_next = PD - 1 + ((g + 1 > KB && g + 1 <= KB + PD) ? 4 : 0);
                constexpr int LG = (g & 1) ? -1 : lg_next - 1 - (g == KB ? 4 : 0);
                static_assert(KB % 2 == 0 || C::F < 4, "pair waits assume an even bias gap");
                constexpr int id0 = g * PPG;
                constexpr int K = PPG == 1 ? (id0 & 7) : -1;
                constexpr int pair = (g * PPG) >> 3;
                if constexpr (g < C::F)
                    gap_stmt<1, LG, true, LOA, K>(Y[g >> 1], wq[g % PD], p_prev[g & 1], la, s_cur[2 * pair], s_cur[2 * pair + 1], gk, gr, pw[pair]);
                else
                    gap_stmt<0, LG, true, LOA, K>(s_next, wq[g % PD], xf[g - C::F], la, s_cur[2 * pair], s_cur[2 * pair + 1], gk, gr, pw[pair]);
                if constexpr (PPG > 1) {   // D < 512: several GELU pieces per gap, as statements of their own
                    [&]<int... Q>(std::integer_sequence<int, Q...>) {
                        ([&] {
                            constexpr int id = g * PPG + Q, pr = id >> 3;
                            gelu_piece<(id & 7)>(s_cur[2 * pr], s_cur[2 * pr + 1], gk, gr, pw[pr]);
                        }(), ...);
                    }(std::make_integer_sequence<int, PPG>{});
                }
                if constexpr (g == KB) bias_init(c + 1, s_next);
                if constexpr (g % DSTEP == DSTEP / 2) {
                    constexpr int j = (g % C::F) / DSTEP;
                    if constexpr (g < C::F) glds16u_j<j>(src_e, dma_voff, dst_e);
                    else glds16u_j<j>(src_m, dma_voff, dst_m);
                }
            }(), ...);
        }(std::make_integer_sequence<int, NG>{});
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        p_out[0] = __builtin_bit_cast(bf16x8, u32x4{pw[0], pw[1], pw[2], pw[3]});
        p_out[1] = __builtin_bit_cast(bf16x8, u32x4{pw[4], pw[5], pw[6], pw[7]});
    };

    bf16x8 pA[2], pB[2];
    pA[0] = pA[1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = c0; c < c1; c += 2) {   // c0 and c1 are even (mlp_fused_plan): no control flow around the accumulators
        iteration(std::integral_constant<int, 0>{}, c, sA, sB, pA, pB);
        iteration(std::integral_constant<int, 1>{}, c + 1, sB, sA, pB, pA);
    }
    // SKIP: the first half of the long-skip operand's rows (k-steps 0 .. F/2-1, natural k order) is requested here, under the
    // tail; the registers of the fc1 input fragments are free from here on
    bf16x8 sk[C::F / 2];
    const bf16_t* sr = nullptr;        // this lane's long-skip row (kept across the first skip phases for the second half's loads:
    if constexpr (SKIP) {              // no address arithmetic -- compiler VALU code -- between the hand-placed MFMAs)
        bool ok_s;
        sr = a.skip + row_of(ok_s) * D + 8 * half_of();
#pragma unroll
        for (int ks = 0; ks < C::F / 2; ++ks) sk[ks] = *reinterpret_cast<const bf16x8*>(sr + 16 * ks);
    }
    // tail: GEMM2 of the last chunk; its block W2(c1-1) (slot 3) was confirmed at the last M barrier and the first PD
    // fragments are already in flight
    [&]<int... FI>(std::integer_sequence<int, FI...>) {
        ([&] {
            constexpr int f = FI, left = C::F - 1 - f, LG = left < PD - 1 ? left : PD - 1;
            constexpr bool RD = f + PD < C::F;
            constexpr int LO = RD ? frag_off(3, f + PD) : 0;
            constexpr int LOA = LO < 65536 ? LO : LO - 65536;
            gap_stmt<1, LG, RD, LOA, -1>(Y[f >> 1], wq[f % PD], pA[f & 1], LO < 65536 ? lds_lo : lds_hi, 0.f, 0.f, gk, gr, pw_none);
        }(), ...);
    }(std::make_integer_sequence<int, C::F>{});
    wait_vmcnt<0>();     // the run-ahead DMA of the padded blocks must not outlive the workgroup's LDS allocation
    // hipcc does not know that the asm statements are MFMAs: left alone it schedules its own reads of the accumulators
    // (v_accvgpr_read, AGPR spills) directly behind the last MFMA, inside its 12-wait-state shadow.  The drain, then one
    // empty asm per tile that "rewrites" it: every compiler read of Y is ordered behind the drain.
    mfma_drain();
#pragma unroll
    for (int t = 0; t < C::NT; ++t) asm volatile("" : "+a"(Y[t]));

    if constexpr (SKIP) {
        // ---- (a) y = Y + b2 as the B fragments of the y half (accumulator k order: the image's y blocks are packed to match),
        //      (b) the accumulators restart from bskip
        __builtin_amdgcn_s_barrier();     // every wave's pieces of the first three skip blocks have landed (wait_vmcnt<0> above)
        {
            const int hs = half_of();
            const float* lb2s = vecs + 4 * D + 4 * hs;
            const float* lbsk = vecs + 6 * D + 4 * hs;
            // TAP (early-exit models: an instantiation of its own -- the stores' address registers would spill inside the product's SKIP phases):
            // y leaves for the next block's output head before skip_linear replaces it
            bool tap_ok = false;
            float* tap = nullptr;
            if constexpr (TAP) { tap = a.y_tap + row_of(tap_ok) * D + 4 * hs; }
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                acc_pin(Y[t]);
                const f32x16 yt = Y[t];
#pragma unroll
                for (int kq = 0; kq < 2; ++kq) {
                    const int ks = 2 * t + kq;
                    unsigned u[4];
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq) {
                        const int col = 16 * ks + 8 * gq, g = 2 * kq + gq;
                        f32x4 q = {yt[4 * g], yt[4 * g + 1], yt[4 * g + 2], yt[4 * g + 3]};
                        q += *reinterpret_cast<const f32x4*>(lb2s + col);
                        if constexpr (TAP) { if (tap_ok) *reinterpret_cast<f32x4*>(tap + col) = q; }
                        u[2 * gq] = pack2(q[0], q[1]);
                        u[2 * gq + 1] = pack2(q[2], q[3]);
                    }
                    xf[ks] = __builtin_bit_cast(bf16x8, u32x4{u[0], u[1], u[2], u[3]});
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(lbsk + 32 * t + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) Y[t][4 * g + e] = bq[e];
                }
                acc_pin(Y[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- 2 NT phases of F MFMAs; stream position 2 nchunks + p = skip block p, in ring slot p & 3 (nchunks is even): blocks
        // 0, 1, 2 were requested by the chunk loop's run-ahead and have landed.  Phase p: B operand and column tile
        //     p <  NT          : y fragments xf[f],            tile p
        //     p <  NT + NT/2   : sk[f % (F/2)] (first half),   tile 2 (p - NT) + (f >= F/2)
        //     else             : xf[f % (F/2)] (second half, requested in the middle of phase NT), tile 2 (p - NT - NT/2) + (f >= F/2)
        // In the middle of phase p: block p+1 landed | barrier | request block p+3 (as the projection phases do).  The 16-byte
        // loads of the second half sit right behind the requests of phase NT: the two waits that follow count them in.
        const char* const wskip = wmlp + (size_t)(2 * a.nchunks) * C::BLK;
        const unsigned lds_lo_s = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
        const unsigned lds_hi_s = lds_lo_s + 65536u;
        constexpr int PDs = C::F / 2 < 8 ? C::F / 2 : 8;
        constexpr int NPs = 2 * C::NT, NGs = NPs * C::F, H2 = C::F / 2;
        bf16x8 ws[PDs];
        [&]<int... J>(std::integer_sequence<int, J...>) {
            (lds_frag<J * 1024>(ws[J], lds_lo_s), ...);
        }(std::make_integer_sequence<int, PDs>{});
        [&]<int... GI>(std::integer_sequence<int, GI...>) {
            ([&] {
                constexpr int g = GI, p = g / C::F, f = g % C::F;
                if constexpr (f == C::F / 2) {
                    // younger than block p+1's requests at this point: block p+2's, plus (p = NT+1, NT+2) the H2 operand loads
                    constexpr int extra = (p == C::NT + 1 || p == C::NT + 2) ? H2 : 0;
                    wait_vmcnt<C::FPW + extra>();
                    __builtin_amdgcn_s_barrier();
                    const char* src = uniform_ptr(wskip + (size_t)(p + 3) * C::BLK);
                    char* dst = smem + ((p + 3) & 3) * C::BLK + (wave * C::FPW) * 1024;
                    [&]<int... J>(std::integer_sequence<int, J...>) { (glds16u_j<J>(src, dma_voff, dst), ...); }(std::make_integer_sequence<int, C::FPW>{});
                    if constexpr (p == C::NT) {       // second half of the long-skip rows -> the registers the y fragments have left
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ks = 0; ks < H2; ++ks) xf[ks] = *reinterpret_cast<const bf16x8*>(sr + 16 * (H2 + ks));
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("" ::: "memory");
                    }
                }
                constexpr int gn = g + PDs, left = NGs - 1 - g;
                constexpr int lg_self = left < PDs - 1 ? left : PDs - 1, lg_next = left - 1 < PDs - 1 ? left - 1 : PDs - 1;
                constexpr int LG = (g & 1) ? -1 : (left >= 1 ? lg_next - (left >= PDs ? 1 : 0) : lg_self);
                constexpr bool RD = gn < NGs;
                constexpr int LO = RD ? ((gn / C::F) & 3) * C::BLK + (gn % C::F) * 1024 : 0;
                constexpr int LOA = LO < 65536 ? LO : LO - 65536;
                constexpr int tile = p < C::NT ? p : 2 * ((p - C::NT) % (C::NT / 2)) + (f >= H2 ? 1 : 0);
                if constexpr (p < C::NT)
                    gap_plain<LG, RD, LOA>(Y[tile], ws[g % PDs], xf[f], LO < 65536 ? lds_lo_s : lds_hi_s);
                else if constexpr (p < C::NT + C::NT / 2)
                    gap_plain<LG, RD, LOA>(Y[tile], ws[g % PDs], sk[f % H2], LO < 65536 ? lds_lo_s : lds_hi_s);
                else
                    gap_plain<LG, RD, LOA>(Y[tile], ws[g % PDs], xf[f % H2], LO < 65536 ? lds_lo_s : lds_hi_s);
            }(), ...);
        }(std::make_integer_sequence<int, NGs>{});
        wait_vmcnt<0>();     // the run-ahead requests of the padded blocks
        mfma_drain();
#pragma unroll
        for (int t = 0; t < C::NT; ++t) asm volatile("" : "+a"(Y[t]));
    }

    // ---- epilogue.  Accumulator layout: lane = token row, register quad g of tile t = columns 32t + 8g + 4h .. +3.
    // (SKIP: the accumulators hold x' = skip_linear([y | skip]) complete with its bias; no bf16 copy of it is needed)
    const int he = half_of();
    if constexpr (PARTIAL) {
        if (wave * 32 >= a.prows) return;          // (this wave's rows lie past the tile: nothing to store, no barrier follows)
        float* pp = a.partial + ((long long)slab * a.prows + wave * 32 + r32) * D + 4 * he;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            const f32x16 yt = Y[t];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(pp + 32 * t + 8 * g) = f32x4{yt[4 * g], yt[4 * g + 1], yt[4 * g + 2], yt[4 * g + 3]};
        }
    } else {
        bool row_ok;
        const long long row = row_of(row_ok);
        if constexpr (!QKV) {
            if (!row_ok) return; // rows past the end of a ragged last tile (both lanes of a row agree; no barrier follows)
        }
        const float* lg_out = vecs + 2 * D + 4 * he;
        const float* lb_out = vecs + 3 * D + 4 * he;
        const float* lb2 = vecs + 4 * D + 4 * he;
        // QKV: every wave runs the phases below, so rows past the end of a ragged tile store into the dump area instead of leaving
        float* xrow = (!QKV || row_ok) ? a.xres + row * D + 4 * he : reinterpret_cast<float*>(reinterpret_cast<char*>(a.qkv_dump) + 8192) + 4 * he;
        bf16_t* orow = (!QKV || row_ok) ? a.out + row * a.ldo + 8 * he : reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(a.qkv_dump) + 12288) + 8 * he;
        f32x4 xl[2][4];   // !LNIN: residual quads of tile t, loaded one tile ahead of the stores (vmcnt retires in order)
        if constexpr (!LNIN) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xl[0][g] = *reinterpret_cast<const f32x4*>(xrow + 8 * g);
        }
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, q4 = {0.f, 0.f, 0.f, 0.f};
        float cshift = 0.f;
#pragma unroll
        for (int t = 0; t < C::NT; ++t) {
            if constexpr (!LNIN) {
                if (t + 1 < C::NT) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) xl[(t + 1) & 1][g] = *reinterpret_cast<const f32x4*>(xrow + 32 * (t + 1) + 8 * g);
                }
            }
            uint2 v[4];
            const f32x16 yt = Y[t];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 q = {yt[4 * g], yt[4 * g + 1], yt[4 * g + 2], yt[4 * g + 3]};
                if constexpr (!SKIP) q += *reinterpret_cast<const f32x4*>(lb2 + 32 * t + 8 * g);
                if constexpr (!LNIN) q = xl[t & 1][g] + q;
                *reinterpret_cast<f32x4*>(xrow + 32 * t + 8 * g) = q;
                v[g] = uint2{pack2(q[0], q[1]), pack2(q[2], q[3])};
                if (t == 0 && g == 0) cshift = q[0];
                const f32x4 dq = q - cshift;
                s4 += dq;
                q4 += dq * dq;
            }
            if (!SKIP && a.out) {
                // bf16 copy as 16-byte row segments: v_permlane32_swap joins the two lane halves (see gemm.hip)
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(v[gp].x, v[gp + 1].x, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(v[gp].y, v[gp + 1].y, false, false);
                    const uint4 o = {s0[0], s1[0], s0[1], s1[1]};
                    *reinterpret_cast<uint4*>(orow + 32 * t + 8 * gp) = o;
                }
            }
            acc_pin(Y[t]);                       // the copy dies here: the LayerNorm pass below re-reads the accumulators instead of a spill slot
            asm volatile("" : "+v"(s4), "+v"(q4));   // the statistics are accumulated HERE (hipcc otherwise sinks them into the ln_out
                                                     // branch below and keeps -- spills -- all 256 row values for it)
            __builtin_amdgcn_sched_barrier(0);   // tile by tile: keeps the live set small (no spills)
        }
        if (QKV || (LNIN && a.ln_out)) {   // LayerNorm of the updated row (the next block's norm1), from the registers (LNIN mode only): stored as
                                           // bf16 -- or (QKV) kept as the B fragments of the qkv phases below (accumulator k order)
            float mean, rstd;
            ln_stats_shifted<D>(cshift, (s4[0] + s4[1]) + (s4[2] + s4[3]), (q4[0] + q4[1]) + (q4[2] + q4[3]), mean, rstd);
            const float shift = -mean * rstd;
            const float* lb2r = lb2;
            asm volatile("" : "+v"(lb2r));        // a second read of the bias quads from LDS, not 64 values carried over from pass 1 (spills)
            // where the 16-byte piece of k-step ks = 2t + gp2/2 goes: row-major rows, or (ln_out_frag) the MFMA fragment order the
            // attention launch loads straight into registers -- [32-row group][k-step][lane] x 16 bytes, a wave's store = 1 KB contiguous
            bf16_t* lnp = nullptr;
            long long lnstride = 16;
            if constexpr (!QKV && !PARTIAL) {
                unsigned l = (unsigned)threadIdx.x;
                asm volatile("" : "+v"(l));
                if (a.ln_out_frag) { lnp = a.ln_out_frag + (((long long)tile_idx * 4 + (l >> 6)) * C::F * 64 + (l & 63)) * 8; lnstride = 512; }
                else lnp = a.ln_out + row * D + 8 * he;
            }
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                uint2 v[4];
                const f32x16 yt = Y[t];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // the updated row is rebuilt from the (read-only) accumulators: cheaper than writing it back in pass 1
                    f32x4 q = {yt[4 * g], yt[4 * g + 1], yt[4 * g + 2], yt[4 * g + 3]};
                    if constexpr (!SKIP) q += *reinterpret_cast<const f32x4*>(lb2r + 32 * t + 8 * g);
                    const f32x4 gv = *reinterpret_cast<const f32x4*>(lg_out + 32 * t + 8 * g), bv = *reinterpret_cast<const f32x4*>(lb_out + 32 * t + 8 * g);
                    const f32x4 w = (q * rstd + shift) * gv + bv;
                    v[g] = uint2{pack2(w[0], w[1]), pack2(w[2], w[3])};
                }
                if constexpr (QKV) {
                    // k-step 2t + kq of the qkv product = quads 2kq, 2kq + 1 of tile t (as norm2 feeds fc1 in the prologue)
                    xf[2 * t] = __builtin_bit_cast(bf16x8, u32x4{v[0].x, v[0].y, v[1].x, v[1].y});
                    xf[2 * t + 1] = __builtin_bit_cast(bf16x8, u32x4{v[2].x, v[2].y, v[3].x, v[3].y});
                    asm volatile("" : "+v"(xf[2 * t]), "+v"(xf[2 * t + 1]));   // computed HERE (hipcc otherwise sinks the arithmetic to the first use
                                                                               // in the phases and keeps -- spills -- every LDS quad loaded for it)
                    acc_pin(Y[t]);
                } else {
#pragma unroll
                    for (int gp2 = 0; gp2 < 4; gp2 += 2) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(v[gp2].x, v[gp2 + 1].x, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(v[gp2].y, v[gp2 + 1].y, false, false);
                        const uint4 o = {s0[0], s1[0], s0[1], s1[1]};
                        *reinterpret_cast<uint4*>(lnp + (2 * t + gp2 / 2) * lnstride) = o;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if constexpr (QKV) {
            // ---- the qkv phases.  The row accumulators are dead from here on (x has been stored above: its stores drain under the
            // first phases -- vmcnt is in order, so the first request that must be confirmed behind them waits for them).  Stream
            // position of qkv block t: behind the MLP (and skip) blocks, slot t & 3; blocks 0..2 were requested by the run-ahead of the
            // loop before and have landed (wait_vmcnt<0> + the barrier here).
            const char* const wqkv = wmlp + ((size_t)2 * a.nchunks + (SKIP ? 2 * C::NT : 0)) * C::BLK;
            const unsigned lds_lo_q = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
            const unsigned lds_hi_q = lds_lo_q + 65536u;
            constexpr int PDq = C::F / 2 < 8 ? C::F / 2 : 8;
            constexpr int NQ = 3 * C::NT, NGq = NQ * C::F;
            // this lane's row in the head-major qkv tensor (dd_internal.h HeadMajor): unit u = t / 2 is Lp rows of 128 bytes further per step;
            // rows past the end of a ragged tile write to a dump area (every wave issues the same number of stores: the waits count them)
            const int pidx = (int)(row / a.tok_l), lidx = (int)(row - (long long)pidx * a.tok_l);
            char* qrow = row_ok ? reinterpret_cast<char*>(a.qkv_out) + (((long long)pidx * (3 * a.hm.H)) * a.hm.Lp + lidx) * 128 + 16 * he
                                : reinterpret_cast<char*>(a.qkv_dump) + (tid * 2) * 16;
            const long long ustride = row_ok ? (long long)a.hm.Lp * 128 : 0;    // bytes between consecutive (q | k | v, head) units of an image
            const int tstride = row_ok ? 64 : 0, gstride = row_ok ? 32 : 16;    // bytes between the two tiles of a unit / the two 16-byte pieces of a tile
            // the x / bf16-copy stores above are in the vector-memory queue in front of everything the phases issue: the counted
            // waits below only ever ask for requests issued INSIDE the phases, so older stores need no accounting
            __builtin_amdgcn_s_barrier();
            bf16x8 wq2[PDq];
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (lds_frag<J * 1024>(wq2[J], lds_lo_q), ...);
            }(std::make_integer_sequence<int, PDq>{});
            f32x16 qa, qb;
#pragma unroll
            for (int e = 0; e < 16; ++e) { qa[e] = 0.f; qb[e] = 0.f; }
            asm volatile("" : "+a"(qa), "+a"(qb));
            // tile tt (in acc) -> bf16, two 16-byte stores per lane (v_permlane32_swap pairs the lane halves), accumulator back to zero
            auto flush_tile = [&](int tt, f32x16& acc) {
                uint2 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = uint2{pack2(acc[4 * g], acc[4 * g + 1]), pack2(acc[4 * g + 2], acc[4 * g + 3])};
                char* dst = qrow + (long long)(tt >> 1) * ustride + (tt & 1) * tstride;
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(v[gp].x, v[gp + 1].x, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(v[gp].y, v[gp + 1].y, false, false);
                    const unsigned o0 = s0[0], o1 = s1[0], o2 = s0[1], o3 = s1[1];
                    *reinterpret_cast<uint4*>(dst + (gp >> 1) * gstride) = uint4{o0, o1, o2, o3};
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            };
            [&]<int... GI>(std::integer_sequence<int, GI...>) {
                ([&] {
                    constexpr int g = GI, t = g / C::F, f = g % C::F;
                    if constexpr (f == C::F / 2) {
                        // younger than block t+1's requests (issued in the middle of phase t-2): the two stores of phase t-2 (tile t-3, if
                        // any), block t+2's requests, the two stores of phase t-1 (tile t-2, if any)
                        constexpr int younger = C::FPW + (t - 2 >= 1 ? 2 : 0) + (t - 1 >= 1 ? 2 : 0);
                        if constexpr (t >= 2) wait_vmcnt<younger>();
                        __builtin_amdgcn_s_barrier();
                        const char* src = uniform_ptr(wqkv + (size_t)(t + 3) * C::BLK);
                        char* dst = smem + ((t + 3) & 3) * C::BLK + (wave * C::FPW) * 1024;
                        [&]<int... J>(std::integer_sequence<int, J...>) { (glds16u_j<J>(src, dma_voff, dst), ...); }(std::make_integer_sequence<int, C::FPW>{});
                    }
                    if constexpr (f == C::F / 2 + 1 && t >= 1) {       // second half of phase t: tile t-1 leaves (its last MFMA is F/2 + 1 gaps old)
                        asm volatile("" ::: "memory");
                        if constexpr ((t - 1) & 1) { asm volatile("" : "+a"(qb)); flush_tile(t - 1, qb); }
                        else { asm volatile("" : "+a"(qa)); flush_tile(t - 1, qa); }
                    }
                    if constexpr (f == C::F - 2 && t >= 1) {           // ... and is zero again before phase t+1 accumulates into it
                        if constexpr ((t - 1) & 1) asm volatile("" : "+a"(qb) :: "memory");
                        else asm volatile("" : "+a"(qa) :: "memory");
                    }
                    constexpr int gn = g + PDq, left = NGq - 1 - g;
                    constexpr int lg_self = left < PDq - 1 ? left : PDq - 1, lg_next = left - 1 < PDq - 1 ? left - 1 : PDq - 1;
                    constexpr int LG = (g & 1) ? -1 : (left >= 1 ? lg_next - (left >= PDq ? 1 : 0) : lg_self);
                    constexpr bool RD = gn < NGq;
                    constexpr int LO = RD ? ((gn / C::F) & 3) * C::BLK + (gn % C::F) * 1024 : 0;
                    constexpr int LOA = LO < 65536 ? LO : LO - 65536;
                    if constexpr (t & 1) gap_plain<LG, RD, LOA>(qb, wq2[g % PDq], xf[f], LO < 65536 ? lds_lo_q : lds_hi_q);
                    else gap_plain<LG, RD, LOA>(qa, wq2[g % PDq], xf[f], LO < 65536 ? lds_lo_q : lds_hi_q);
                }(), ...);
            }(std::make_integer_sequence<int, NGq>{});
            mfma_drain();
            if constexpr ((NQ - 1) & 1) { asm volatile("" : "+a"(qb)); flush_tile(NQ - 1, qb); }
            else { asm volatile("" : "+a"(qa)); flush_tile(NQ - 1, qa); }
            asm volatile("" : "+a"(qa), "+a"(qb));
            wait_vmcnt<0>();     // the run-ahead requests of the padded blocks must not outlive the workgroup's LDS allocation
        }
    }
}

// LNIN = false: the input rows are a.X (bf16, e.g. the output of a LayerNorm kernel) and the residual x is read in the
//               epilogue;
// LNIN = true:  the kernel reads the fp32 residual rows x ONCE, straight into the output accumulators (x is then already
//               part of Y: no second read), computes LayerNorm(x) * gamma + beta in registers (reference
//               models/uvit.py:207 norm2) and converts the normalised accumulator registers in place into the MFMA
//               B fragments of fc1 -- their k order is the accumulator order, the W1 image is packed to match.
// Either way an optional second LayerNorm (the NEXT block's norm1, models/uvit.py:206) of the updated rows is written
// as bf16 from the epilogue (a.ln_out), so neither LayerNorm of a block needs a launch or an HBM round trip of x.
// Workgroups [0, tiles_main) take a main tile each; the rest are the hidden-split workgroups of the extra-token tiles.
template <int D, bool LNIN, bool PROJ, bool SKIP = false, bool QKV = false, bool TAP = false>
__global__ void __launch_bounds__(256) mlp_fused_kernel(const MlpFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < a.tiles_main) {
        mlp_body<D, LNIN, false, PROJ, SKIP, QKV, TAP>(a, smem, blockIdx.x, 0, a.nchunks, 0);
    } else {
        const int e = blockIdx.x - a.tiles_main;
        const int lt = e / a.groups, g = e - lt * a.groups;
        const int c0 = g * a.cpg;
        mlp_body<D, LNIN, true, PROJ>(a, smem, lt, c0, c0 + a.cpg < a.nchunks ? c0 + a.cpg : a.nchunks, e);
    }
}

// Extra-token rows: x[row] += b2 + sum over the groups' partial slabs (fixed order), the optional bf16 copy, and the
// optional LayerNorm of the updated row (the next block's norm1).  One wave per row, VPL = D / 64 columns per lane.
template <int VPL>
__global__ void __launch_bounds__(256) mlp_reduce_kernel(const MlpFusedArgs a) {
    constexpr int D = VPL * 64;
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);      // logical extra row
    if (r >= a.n_extra) return;
    const int lt = (int)(r / a.prows), rr = (int)(r % a.prows);
    const long long b = r / a.tok_e;
    const long long row = b * a.tok_l + (r - b * a.tok_e);
    const int col = lane * VPL;
    float acc[VPL], xv[VPL];
#pragma unroll
    for (int e = 0; e < VPL; ++e) acc[e] = a.b2[col + e];
    // every slab row (and the residual row) is requested before the first add: one memory round trip, not one per group;
    // the sum still runs over the groups in ascending order
    float* xp = a.xres + row * D + col;
    float pv[kGroups][VPL];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        const float* pp = a.partial + (((long long)lt * a.groups + (g < a.groups ? g : 0)) * a.prows + rr) * D + col;
#pragma unroll
        for (int e = 0; e < VPL; ++e) pv[g][e] = pp[e];
    }
#pragma unroll
    for (int e = 0; e < VPL; ++e) xv[e] = xp[e];
    float gv[VPL], bv[VPL];
    if (a.ln_out) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) { gv[e] = a.ln_out_g[col + e]; bv[e] = a.ln_out_b[col + e]; }
    }
#pragma unroll
    for (int g = 0; g < kGroups; ++g)
        if (g < a.groups) {
#pragma unroll
            for (int e = 0; e < VPL; ++e) acc[e] += pv[g][e];
        }
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < VPL; ++e) { xv[e] = a.reduce_set ? acc[e] : xv[e] + acc[e]; xp[e] = xv[e]; sum += xv[e]; }
    if (a.y_tap) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) a.y_tap[row * D + col + e] = xv[e];
    }
    if (a.out) {
#pragma unroll
        for (int e = 0; e < VPL; ++e) a.out[row * a.ldo + col + e] = f2bf(xv[e]);
    }
    if (a.ln_out) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum / (float)D;
        float q2 = 0.f;
#pragma unroll
        for (int e = 0; e < VPL; ++e) { const float d = xv[e] - mean; q2 += d * d; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) q2 += __shfl_xor(q2, o);
        const float rstd = 1.0f / sqrtf(q2 / (float)D + 1e-5f);
#pragma unroll
        for (int e = 0; e < VPL; ++e)
            a.ln_out[row * D + col + e] = f2bf((xv[e] - mean) * rstd * gv[e] + bv[e]);
    }
}

// The extra-token rows of a SKIP launch (they left the fused launch through the hidden-split workgroups and the reduce
// kernel, which stored y as bf16 in a.out): x' = [y | skip] . Wskip^T + bskip, stored fp32, and norm1(x') as bf16 --
// what the main tiles do in their SKIP phases.  The kernel is pure latency (B * extras rows against 2 D^2 weights), so it is
// cut for depth, not throughput: one workgroup per 32 rows with ONE WAVE PER COLUMN TILE (D/32 waves: 64 MFMAs each at
// D = 512, as two chains), the rows' y and skip operands parked once in LDS (padded rows: conflict-free fragment reads; the
// y half is read in accumulator k order as two 8-byte pieces per k-step), the weights straight from the image's fragment
// order, 16 fragments in flight per wave, LayerNorm statistics (two-pass) exchanged through LDS.
// WPG = waves (= column tiles) per workgroup.  WPG == D / 32 with LN: the form above.  WPG < D / 32 without LN (grid.y = D / 32 / WPG
// column groups): the product's form where the consumer normalises the rows itself (qkv_attention_kernel) -- a workgroup then reads
// WPG x 2 D x 64 bytes of weights instead of all 2 D^2 x 2, which is what the launch's time is (a CU's L2 port: 1 MB ~ 15 us).
template <int D, int WPG = D / 32, bool LN = true>
__global__ void __launch_bounds__(WPG * 64) skip_rows_ln_kernel(const MlpFusedArgs a) {
    using C = MlpCfg<D>;
    static_assert(!LN || WPG == D / 32, "LayerNorm needs the whole row in one workgroup");
    constexpr int H2 = C::F / 2, PITCH = D * 2 + 16, NW = C::NT, NTHR = WPG * 64;
    extern __shared__ __attribute__((aligned(16))) char srl[];
    char* ybuf = srl;                       // [32][PITCH]
    char* sbuf = srl + 32 * PITCH;          // [32][PITCH]
    float* red = reinterpret_cast<float*>(srl + 64 * PITCH);   // [2][NW][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r32 = lane & 31;
    auto row_of = [&](int i) -> long long {
        const int q = i < a.n_extra ? i : a.n_extra - 1, b = q / a.tok_e;
        return (long long)b * a.tok_l + (q - b * a.tok_e);
    };
    // ---- park the operand rows: 32 rows x D bf16 each, 16-byte chunks
    constexpr int CPR = D / 8, ITEMS = 2 * 32 * CPR, PER_ALL = (ITEMS + NTHR - 1) / NTHR, PER = PER_ALL < 8 ? PER_ALL : 8, ROUNDS = (PER_ALL + PER - 1) / PER;
    f32x4 stg[PER];
    auto stage_load = [&](int rd) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int it = tid + (rd * PER + k) * NTHR, which = it / (32 * CPR), rem = it % (32 * CPR), r = rem / CPR, ch = rem % CPR;
            const long long row = row_of(blockIdx.x * 32 + r);
            const bf16_t* src = which == 0 ? a.out + row * a.ldo : a.skip + row * D;
            stg[k] = it < ITEMS ? *reinterpret_cast<const f32x4*>(src + ch * 8) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stage_store = [&](int rd) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int it = tid + (rd * PER + k) * NTHR, which = it / (32 * CPR), rem = it % (32 * CPR), r = rem / CPR, ch = rem % CPR;
            if (it < ITEMS) *reinterpret_cast<f32x4*>((which == 0 ? ybuf : sbuf) + r * PITCH + ch * 16) = stg[k];
        }
    };
    constexpr bool DMA_PARK = !LN && D == 512;     // a row = exactly one 1 KB LDS-DMA piece: all 64 pieces of the workgroup in flight at once
    if constexpr (DMA_PARK) {                      // (the register-staged rounds below are a chain of memory round trips at 128 threads)
#pragma unroll
        for (int i = 0; i < 32 / WPG; ++i) {
            const int r = wave + i * WPG;
            const long long row = row_of(blockIdx.x * 32 + r);
            glds16(a.out + row * a.ldo + lane * 8, ybuf + r * PITCH);
            glds16(a.skip + row * D + lane * 8, sbuf + r * PITCH);
        }
    } else {
        stage_load(0);
    }
    // this wave's column tile: weights of the y half (block t), first fragments requested before the operands are parked
    const int t = blockIdx.y * WPG + wave;
    const char* const wskip = a.wimg + ((size_t)a.nproj + 2 * (size_t)a.nchunks) * C::BLK;
    const bf16x8* wy = reinterpret_cast<const bf16x8*>(wskip + (size_t)t * C::BLK) + lane;
    const bf16x8* ws0 = reinterpret_cast<const bf16x8*>(wskip + (size_t)(C::NT + t / 2) * C::BLK) + (t & 1) * H2 * 64 + lane;
    const bf16x8* ws1 = reinterpret_cast<const bf16x8*>(wskip + (size_t)(C::NT + C::NT / 2 + t / 2) * C::BLK) + (t & 1) * H2 * 64 + lane;
    // weight fragment kk of this tile's 2 F k-steps: the y half (block t), then the two passes of the skip half
    auto wfrag = [&](int kk) -> bf16x8 {
        return kk < C::F ? wy[kk * 64] : kk < C::F + H2 ? ws0[(kk - C::F) * 64] : ws1[(kk - C::F - H2) * 64];
    };
    constexpr int G = 16;                         // fragments in flight: a rolling window, refilled as it is consumed
    bf16x8 wf[G];
#pragma unroll
    for (int f = 0; f < G; ++f) wf[f] = wfrag(f);
    if constexpr (DMA_PARK) {
        wait_vmcnt<0>();
    } else {
        stage_store(0);
#pragma unroll
        for (int rd = 1; rd < ROUNDS; ++rd) { stage_load(rd); stage_store(rd); }
    }
    __syncthreads();
    f32x16 acc0, acc1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(a.bskip + 32 * t + 8 * g + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc0[4 * g + e] = bq[e]; acc1[4 * g + e] = 0.f; }
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const char* yrow = ybuf + r32 * PITCH + 8 * h;       // accumulator k order: columns 16 ks + 4 h .. +3 and 16 ks + 8 + 4 h .. +3
    const char* srow = sbuf + r32 * PITCH + 16 * h;      // natural k order: columns 16 ks + 8 h .. +7
#pragma unroll
    for (int kk = 0; kk < 2 * C::F; ++kk) {
        bf16x8 bfr;
        if (kk < C::F) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(yrow + 32 * kk), hi = *reinterpret_cast<const u32x2*>(yrow + 32 * kk + 16);
            bfr = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        } else {
            bfr = *reinterpret_cast<const bf16x8*>(srow + 32 * (kk - C::F));
        }
        if (kk & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk % G], bfr, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[kk % G], bfr, acc0, 0, 0, 0);
        if (kk + G < 2 * C::F) wf[kk % G] = wfrag(kk + G);
    }
    const f32x16 acc = acc0 + acc1;
    if constexpr (!LN) {
        const int idx = blockIdx.x * 32 + r32;
        if (idx >= a.n_extra) return;
        const long long row = row_of(idx);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(a.xres + row * D + 32 * t + 8 * g + 4 * h) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        return;
    }
    // two-pass LayerNorm statistics over the row's D columns: lane halves via shuffle, the waves via LDS
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s1 += acc[e];
    s1 += __shfl_xor(s1, 32);
    if (h == 0) red[wave * 32 + r32] = s1;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w * 32 + r32];
    const float mean = tot / (float)D;
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = acc[e] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32);
    if (h == 0) red[(NW + wave) * 32 + r32] = s2;
    __syncthreads();
    float tq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tq += red[(NW + w) * 32 + r32];
    const float rstd = 1.0f / sqrtf(tq / (float)D + 1e-5f);
    const int idx = blockIdx.x * 32 + r32;
    if (idx >= a.n_extra) return;
    const long long row = row_of(idx);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = 32 * t + 8 * g + 4 * h;
        const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4*>(a.xres + row * D + col) = v;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(a.ln_out_g + col), bv = *reinterpret_cast<const f32x4*>(a.ln_out_b + col);
        const f32x4 w = (v - mean) * rstd * gv + bv;
        *reinterpret_cast<uint2*>(a.ln_out + row * D + col) = uint2{pack2(w[0], w[1]), pack2(w[2], w[3])};
    }
}

// qkv of the extra-token rows of a QKV launch: out[row, 32t .. 32t+31] = norm1(x)[row] . Wqkv[32t ..]^T from the bf16 LayerNorm rows
// the reduce / skip_rows launch wrote (a.ln_out), head-major stores.  Workgroup t = one 32-column tile of 128 rows (a 32-row
// group per wave), like proj_rows_kernel: every load up front, whole rows into a wave-private LDS strip, weights from the
// image's fragment order (accumulator k order: the row fragments are gathered as two 8-byte pieces per k-step).
template <int D>
__global__ void __launch_bounds__(256) qkv_rows_kernel(const MlpFusedArgs a) {
    using C = MlpCfg<D>;
    constexpr int LPR = D / 8, RPI = 64 / LPR, NI = 32 / RPI, IPITCH = 1024 + 32;
    extern __shared__ __attribute__((aligned(16))) char strip_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
    const int t = blockIdx.x, idx0 = blockIdx.y * 128 + wave * 32;
    char* strip = strip_lds + wave * (NI * IPITCH);
    auto row_of = [&](int idx) -> long long {
        const int q = idx < a.n_extra ? idx : a.n_extra - 1, b = q / a.tok_e;
        return (long long)b * a.tok_l + (q - b * a.tok_e);
    };
#pragma unroll
    for (int i = 0; i < NI; ++i)
        glds16(a.ln_out + row_of(idx0 + i * RPI + lane / LPR) * D + (lane % LPR) * 8, strip + i * IPITCH);
    __builtin_amdgcn_sched_barrier(0);
    const int idx = idx0 + (lane & 31);
    const bool ok = idx < a.n_extra;
    const long long row = row_of(idx);
    const size_t pos = (size_t)a.nproj + 2 * (size_t)a.nchunks + (size_t)a.nskip;
    const bf16x8* wb = reinterpret_cast<const bf16x8*>(a.wimg + (pos + t) * C::BLK) + lane;
    bf16x8 wf[C::F];
#pragma unroll
    for (int ks = 0; ks < C::F; ++ks) wf[ks] = wb[ks * 64];
    wait_vmcnt<0>();
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const char* my = strip + ((lane & 31) / RPI) * IPITCH + ((lane & 31) % RPI) * (D * 2) + 8 * h;
    f32x16 acc, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < C::F; ks += 2) {
        const u32x2 l0 = *reinterpret_cast<const u32x2*>(my + 32 * ks), h0 = *reinterpret_cast<const u32x2*>(my + 32 * ks + 16);
        const u32x2 l1 = *reinterpret_cast<const u32x2*>(my + 32 * ks + 32), h1 = *reinterpret_cast<const u32x2*>(my + 32 * ks + 48);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], __builtin_bit_cast(bf16x8, u32x4{l0[0], l0[1], h0[0], h0[1]}), acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks + 1], __builtin_bit_cast(bf16x8, u32x4{l1[0], l1[1], h1[0], h1[1]}), acc1, 0, 0, 0);
    }
    acc += acc1;
    if (!ok) return;
    bf16_t* dst = a.qkv_out + hm_offset(a.hm, (int)row, 32 * t) + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<uint2*>(dst + 8 * g) = uint2{pack2(acc[4 * g], acc[4 * g + 1]), pack2(acc[4 * g + 2], acc[4 * g + 3])};
}

template <int D>
hipError_t launch_d(const MlpFusedArgs& a, hipStream_t s) {
    const size_t lds = MlpCfg<D>::RING + (size_t)(a.nchunks + 1) * 32 * sizeof(float) + 7 * D * sizeof(float);   // ring | bias table (+ one chunk: bias_init(c1) is read, unused) | 7 column vectors
    const int grid = a.tiles_main + a.tiles_left * a.groups;
    if (a.nskip > 0 && (a.nproj <= 0 || a.nskip != D / 16 || !a.skip || !a.bskip || (!a.ln_out && a.nqkv <= 0) || a.nchunks % 2)) return hipErrorInvalidValue;
    if (a.nqkv > 0 && a.nproj <= 0) return hipErrorInvalidValue;
    if (a.y_tap && (a.nskip <= 0 || a.nqkv > 0)) return hipErrorInvalidValue;      // (the tap rides on the SKIP launches of early-exit models)
    if (a.ln_out_frag && (!a.ln_out || a.nqkv > 0 || a.tok_n % 32)) return hipErrorInvalidValue;   // (whole 32-row groups of patch rows per wave)
    if (a.nproj > 0) {
        if constexpr (D % 128 == 0) {
            if (!a.ln_in_g || !a.ao || !a.bproj || a.nproj != D / 32) return hipErrorInvalidValue;
            if (a.nqkv > 0) {
                if (a.nqkv != 3 * D / 32 || !a.qkv_out || !a.qkv_dump || !a.ln_out_g || !a.ln_out_b || a.hm.L != a.tok_l || a.nchunks % 2) return hipErrorInvalidValue;
                if (a.nskip > 0) hipLaunchKernelGGL((mlp_fused_kernel<D, true, true, true, true>), dim3(grid), dim3(256), lds, s, a);
                else hipLaunchKernelGGL((mlp_fused_kernel<D, true, true, false, true>), dim3(grid), dim3(256), lds, s, a);
            } else if (a.nskip > 0 && a.y_tap) hipLaunchKernelGGL((mlp_fused_kernel<D, true, true, true, false, true>), dim3(grid), dim3(256), lds, s, a);
            else if (a.nskip > 0) hipLaunchKernelGGL((mlp_fused_kernel<D, true, true, true>), dim3(grid), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((mlp_fused_kernel<D, true, true>), dim3(grid), dim3(256), lds, s, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else if (a.ln_in_g) {
        hipLaunchKernelGGL((mlp_fused_kernel<D, true, false>), dim3(grid), dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL((mlp_fused_kernel<D, false, false>), dim3(grid), dim3(256), lds, s, a);
    }
    return hipGetLastError();
}

template <int D>
hipError_t launch_reduce_d(const MlpFusedArgs& a, hipStream_t s) {
    if (a.tiles_left <= 0) return hipSuccess;
    hipLaunchKernelGGL(mlp_reduce_kernel<D / 64>, dim3((unsigned)((a.n_extra + 3) / 4)), dim3(256), 0, s, a);
    return hipGetLastError();
}


}  // namespace

static size_t skip_rows_lds(int D) { return (size_t)64 * (D * 2 + 16) + (size_t)2 * (D / 32) * 32 * sizeof(float); }

bool mlp_fused_supported(int D, int hidden) {
    return (D == 64 || D == 128 || D == 256 || D == 512) && hidden % 64 == 0 && hidden >= 64 && hidden <= kMaxHidden;
}

// + four blocks: the kernel's DMA runs up to three blocks past the last chunk (branch-free pipeline); never used as data
size_t mlp_fused_image_bytes(int D, int hidden, bool with_proj, bool with_skip, bool with_qkv) {
    return ((size_t)(hidden / 32 + 2) * 2 + (with_proj ? D / 32 : 0) + (with_skip ? D / 16 : 0) + (with_qkv ? 3 * D / 32 : 0)) * (D / 16) * 1024;
}

void mlp_fused_pack_rows(int D, int nrows, const float* w, unsigned short (*to_bf16)(float), unsigned short* img) {
    const int F = D / 16;
    for (int t = 0; t < nrows / 32; ++t)
        for (int f = 0; f < F; ++f)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, h = lane >> 5;
                for (int j = 0; j < 8; ++j)
                    img[((size_t)t * F + f) * 512 + lane * 8 + j] = to_bf16(w[(size_t)(32 * t + r) * D + 16 * f + 8 * (j >> 2) + 4 * h + (j & 3)]);
            }
}

// skip_linear weight [D, 2D] (nn.Linear layout; input = cat([x, skip]), reference models/uvit.py:199) -> the 2 NT blocks the
// SKIP phases stream (they follow the MLP blocks in the image):
//   block p < NT:           output columns 32p .. 32p+31, fragment f = k-step f of the x half, k index in accumulator order
//                           (its B operand is the block output converted from the accumulators)
//   block NT + j, j < NT/2: fragments 0 .. F/2-1 = column tile 2j, F/2 .. F-1 = tile 2j+1; k-steps 0 .. F/2-1 of the skip half,
//                           natural k order (its B operand is the long-skip tensor's rows as loaded)
//   block NT + NT/2 + j:    the same tiles, k-steps F/2 .. F-1 of the skip half
void mlp_fused_pack_skip(int D, const float* ws, unsigned short (*to_bf16)(float), unsigned short* img) {
    const int F = D / 16, NT = D / 32, H2 = F / 2;
    for (int p = 0; p < 2 * NT; ++p)
        for (int f = 0; f < F; ++f)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 31, h = lane >> 5;
                for (int j = 0; j < 8; ++j) {
                    int row, k;
                    if (p < NT) {
                        row = 32 * p + r;
                        k = 16 * f + 8 * (j >> 2) + 4 * h + (j & 3);
                    } else {
                        const int q = p - NT, pass = q / (NT / 2), jj = q % (NT / 2);
                        row = 32 * (2 * jj + (f >= H2 ? 1 : 0)) + r;
                        k = D + 16 * (pass * H2 + f % H2) + 8 * h + j;
                    }
                    img[((size_t)p * F + f) * 512 + lane * 8 + j] = to_bf16(ws[(size_t)row * 2 * D + k]);
                }
            }
}

// Wproj [D, D] (nn.Linear layout) -> D/32 blocks in front of the MLP image: block t = output columns 32t .. 32t+31 as the
// MFMA A operand, fragment ks = k-step ks in natural k order (its B operand is the attention output as loaded)
void mlp_fused_pack_proj(int D, const float* wp, unsigned short (*to_bf16)(float), unsigned short* img) {
    const int F = D / 16;
    for (int t = 0; t < D / 32; ++t)
        for (int ks = 0; ks < F; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    img[((size_t)t * F + ks) * 512 + lane * 8 + j] = to_bf16(wp[(size_t)(32 * t + (lane & 31)) * D + 16 * ks + 8 * (lane >> 5) + j]);
}

// Row plan (see the header): patch rows in 128-row main tiles, extra rows in tiles split `groups` ways along hidden.
// The split is a function of the hidden size alone -- never of the batch -- so results do not depend on the batch size.
void mlp_fused_plan(int B, int n_patches, int extras, int seq_len, int hidden, MlpFusedArgs& a) {
    const int nchunks = hidden / 32;
    a.nchunks = nchunks;
    a.tok_n = n_patches; a.tok_e = extras; a.tok_l = seq_len;
    a.n_main = B * n_patches;
    a.n_extra = B * extras;
    a.tiles_main = (a.n_main + 127) / 128;
    // The hidden-split tiles run behind the main ones with the whole chip free, and their cost is what ONE workgroup moves through
    // its CU's L2 port (rows in, weights, slab out): 32-row tiles (one wave's rows; the other three waves share the weight DMA and
    // idle along) quarter the row and slab share of that -- 4 x the workgroups cost nothing there.
    a.prows = 32;
    a.tiles_left = (a.n_extra + a.prows - 1) / a.prows;
    int g = nchunks / 2 < kGroups ? nchunks / 2 : kGroups;                   // >= 2 chunks per group (the kernel unrolls by 2)
    if (g < 1) g = 1;
    a.cpg = ((nchunks + g - 1) / g + 1) & ~1;
    a.groups = (nchunks + a.cpg - 1) / a.cpg;
}

size_t mlp_fused_partial_bytes(int max_batch, int extras, int D, int hidden) {
    MlpFusedArgs a{};
    mlp_fused_plan(max_batch, 1, extras, 1 + extras, hidden, a);
    return (size_t)a.tiles_left * a.groups * a.prows * D * sizeof(float);
}

// Host: nn.Linear weights (fp32, [out, in]) -> the fragment-ordered bf16 image the kernel streams + permuted fc1 bias.
// kperm: W1's k index inside every group of 16 is permuted to the accumulator order (the kernel's LNIN mode builds the
// fc1 B fragments from accumulator registers); W2's k index always is.
void mlp_fused_pack(int D, int hidden, const float* w1, const float* b1, const float* w2, bool kperm,
                    unsigned short (*to_bf16)(float), unsigned short* img, float* b1p) {
    const int F = D / 16, NT = D / 32, KS = D / 16, nchunks = hidden / 32;
    for (int c = 0; c < nchunks; ++c) {
        unsigned short* blk1 = img + (size_t)(2 * c) * F * 512;
        unsigned short* blk2 = img + (size_t)(2 * c + 1) * F * 512;
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 31, h = lane >> 5;
            for (int ks = 0; ks < KS; ++ks)
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * ks + (kperm ? 8 * (j >> 2) + 4 * h + (j & 3) : 8 * h + j);
                    blk1[(size_t)ks * 512 + lane * 8 + j] = to_bf16(w1[(size_t)(32 * c + r) * D + k]);
                }
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
    const int bias = (kMaxHidden + 32 + 7 * 512) * (int)sizeof(float);
#define DD_ATTR(DV)                                                                                          \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                MlpCfg<DV>::RING + bias);                                                    \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                MlpCfg<DV>::RING + bias);                                                    \
    if constexpr (DV % 128 == 0) {                                                                           \
        if (e == hipSuccess)                                                                                 \
            e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    MlpCfg<DV>::RING + bias);                                                \
        if (e == hipSuccess)                                                                                 \
            e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    MlpCfg<DV>::RING + bias);                                                \
        if (e == hipSuccess)                                                                                 \
            e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV, true, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    MlpCfg<DV>::RING + bias);                                                \
        if (e == hipSuccess)                                                                                 \
            e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    MlpCfg<DV>::RING + bias);                                                \
        if (e == hipSuccess)                                                                                 \
            e = hipFuncSetAttribute((const void*)mlp_fused_kernel<DV, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    MlpCfg<DV>::RING + bias);                                                \
    }
    DD_ATTR(64) DD_ATTR(128) DD_ATTR(256) DD_ATTR(512)
#undef DD_ATTR
#define DD_ATTR_P(DV)                                                                                        \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void*)qkv_rows_kernel<DV>, hipFuncAttributeMaxDynamicSharedMemorySize, kProjRowsLds); \
    if (e == hipSuccess)                                                                                     \
        e = hipFuncSetAttribute((const void*)skip_rows_ln_kernel<DV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)skip_rows_lds(DV));
    DD_ATTR_P(128) DD_ATTR_P(256) DD_ATTR_P(512)
#undef DD_ATTR_P
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)skip_rows_ln_kernel<512, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)skip_rows_lds(512));
    return e;
}

// skip_linear + norm1 of the extra-token rows of a SKIP launch (after launch_mlp_reduce, which stored their y in a.out); no-op without extras
hipError_t launch_skip_rows_ln(const MlpFusedArgs& a, int D, hipStream_t s, bool with_ln) {
    if (a.n_extra <= 0 || a.nskip <= 0) return hipSuccess;
    if (!a.out || !a.skip || !a.bskip || (with_ln && (!a.ln_out || !a.ln_out_g || !a.ln_out_b))) return hipErrorInvalidValue;
    const size_t lds = skip_rows_lds(D);
    if (!with_ln) {     // column-split form: 2 column tiles per workgroup (the consumer normalises the rows)
        if (D != 512) return hipErrorInvalidValue;
        hipLaunchKernelGGL((skip_rows_ln_kernel<512, 2, false>), dim3((a.n_extra + 31) / 32, 512 / 32 / 2), dim3(128), lds, s, a);
        return hipGetLastError();
    }
    const dim3 grid((a.n_extra + 31) / 32);
    switch (D) {
        case 128: hipLaunchKernelGGL((skip_rows_ln_kernel<128>), grid, dim3(256), lds, s, a); break;
        case 256: hipLaunchKernelGGL((skip_rows_ln_kernel<256>), grid, dim3(512), lds, s, a); break;
        case 512: hipLaunchKernelGGL((skip_rows_ln_kernel<512>), grid, dim3(1024), lds, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_qkv_rows(const MlpFusedArgs& a, int D, hipStream_t s) {
    if (a.n_extra <= 0 || a.nqkv <= 0) return hipSuccess;
    if (!a.ln_out || !a.qkv_out || a.nqkv != 3 * D / 32) return hipErrorInvalidValue;
    const dim3 grid(a.nqkv, (a.n_extra + 127) / 128);
    switch (D) {
        case 128: hipLaunchKernelGGL((qkv_rows_kernel<128>), grid, dim3(256), kProjRowsLds, s, a); break;
        case 256: hipLaunchKernelGGL((qkv_rows_kernel<256>), grid, dim3(256), kProjRowsLds, s, a); break;
        case 512: hipLaunchKernelGGL((qkv_rows_kernel<512>), grid, dim3(256), kProjRowsLds, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// the fused launch itself; the extra-token rows are finished by launch_mlp_reduce right behind it
hipError_t launch_mlp_fused(const MlpFusedArgs& a, int D, hipStream_t s) {
    switch (D) {
        case 64: return launch_d<64>(a, s);
        case 128: return launch_d<128>(a, s);
        case 256: return launch_d<256>(a, s);
        case 512: return launch_d<512>(a, s);
    }
    return hipErrorInvalidValue;
}
hipError_t launch_mlp_reduce(const MlpFusedArgs& a, int D, hipStream_t s) {
    switch (D) {
        case 64: return launch_reduce_d<64>(a, s);
        case 128: return launch_reduce_d<128>(a, s);
        case 256: return launch_reduce_d<256>(a, s);
        case 512: return launch_reduce_d<512>(a, s);
        case 768: return launch_reduce_d<768>(a, s);     // the K-split extra-token tiles of rowlin.hip
    }
    return hipErrorInvalidValue;
}

}  // namespace dd
