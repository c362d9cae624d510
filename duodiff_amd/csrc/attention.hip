// Single-pass attention for the U-ViT token sequence (L <= 288, head_dim = 64) on gfx950.
//
// reference models/uvit.py:155-164: q,k,v = split(qkv) ; softmax(q k^T / sqrt(64)) v per (b, h),
// fp32 softmax, no mask, heads merged back as "B H L D -> B L (H D)".
//
// Input: the qkv Linear's output in HEAD-MAJOR order (dd_internal.h HeadMajor: the q, k and v rows of one (image, head) are
// Lp contiguous 128-byte rows each), so a workgroup's K and V are two linear 33 KB reads -- staged by LDS-DMA in bf16 mode
// (no registers, 17 instructions per wave) -- instead of 128-byte pieces at a 3 KB stride.
// One workgroup = one (image, head).  K ([L,64], row-major) and V of
// the head live in LDS for the whole kernel; every wave takes 32-query chunks:
//   S^T[key, q] = K . Q^T        (MFMA A = K rows from LDS, B = Q fragment straight from HBM)
//   softmax over keys            = over accumulator registers (+ one 32-lane exchange): the key
//                                  index sits in the 16 registers x 2 lane-halves of each 32x32
//                                  tile and the query on the lane, so no LDS round trip
//   O^T[d, q]  = V^T . P^T       (A = V^T from LDS, B = the S^T accumulators re-used in place as
//                                  the next MFMA's operand: their k order is the accumulator row
//                                  order  row = (e&3) + 8*(e>>2) + 4*(lane>>5))
// The L x L score matrix is never written anywhere (reference: 270 MB per layer at B=128).
// Works in bf16 (v_mfma_f32_32x32x16_bf16) and in the fp32 parity mode (v_mfma_f32_32x32x2_f32).
#include "dd_internal.h"

#include <type_traits>
#include <utility>

namespace dd {
namespace {

constexpr int kMaxKeyTiles = 9;          // 9 x 32 = 288 >= 258
constexpr int kLP = kMaxKeyTiles * 32;   // padded key count held in LDS
constexpr int kHD = 64;
constexpr int kPartBytes = 4 * 2 * 66 * 4;   // split last query chunk: [4 waves][2 queries][64 d, max, sum] fp32

__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (round to nearest even, as f2bf)
    typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
    typedef float f32v2 __attribute__((ext_vector_type(2)));
    const f32v2 q = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16v2));
}

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

// XOR swizzles of the bf16 K / V images in LDS (128-byte rows of eight 16-byte chunks; chunk ch of row r is stored at chunk ch ^ swz(r)).
// Each serves the image's READ pattern and its WRITE pattern without a bank conflict (MI355X_MICROARCH.md, LDS: a ds_read_b128 serves 4 groups of
// 16 lanes over 64 banks = two rows' worth, a ds_write_b128 8 groups of 8 contiguous lanes over 32 banks = one row's worth):
//   K  read: a lane group holds 8 even + 8 odd rows with 8 different (r >> 1) & 7 each; write (qkv_attention_kernel's epilogue): 8 consecutive
//            rows -- (r >> 1) & 7 alone gives them only 4 different chunks (2-way), the r & 1 bit in the top position makes it 8;
//   V  read: ds_read_b64_tr_b16 of a 4-row block by 32 lanes, rows r & 3 = 0..3 at a fixed (r >> 2) & 1; write: 8 consecutive rows.
__device__ __forceinline__ int k_swz(int r) { return ((r >> 1) & 7) ^ ((r & 1) << 2); }
__device__ __forceinline__ int v_swz(int r) { return (2 * (r & 3)) ^ ((r >> 2) & 1); }

template <typename T> struct AttnLayout;
template <> struct AttnLayout<bf16_t> {
    // K rows are 128 B, LDS-DMA'd linearly (a 1 KB piece = 8 rows); 16-byte chunk ch of row r sits at slot ch ^ k_swz(r)
    // (the swizzle is applied to the SOURCE address of the DMA and again on the read: conflict-free ds_read_b128 fragments,
    // the same image gemm.hip uses)
    static constexpr int kRowK = 128;
    // V stays ROW-major ([key][64 d], 128-byte rows, 16-byte chunk ch of row r stored at chunk ch ^ v_swz(r)): LDS-DMA'd like
    // K, and read as the V^T MFMA operand with the transposing ds_read_b64_tr_b16 (a block of 4 keys x
    // 16 d per 16 lanes; the XOR puts the four rows of a block into the four bank quarters: conflict-free).  The first
    // version wrote V transposed with 4-byte scattered LDS writes -- 12 of the kernel's 47 us.
    static constexpr int kRowV = 128;
    static constexpr int kVBytes = kLP * kRowV;
};
template <> struct AttnLayout<float> {
    static constexpr int kRowK = 256 + 16;        // 68 dwords = 4 mod 64
    static constexpr int kRowV = kLP * 4 + 16;    // V^T row (fp32 path keeps the transposed image): 292 dwords = 4 mod 64
    static constexpr int kVBytes = kHD * kRowV;
};

// One 32-query chunk against NT key tiles tile(0..NT-1) (a negative index = skip): unnormalised
// O^T (two 32(d) x 32(q) tiles), the chunk's running max and the sum of exponentials.
template <typename T, int NKT, int NT, bool KPERM, typename TileFn>
__device__ __forceinline__ void attend_tiles(const char* Ks, const char* Vt, int L, int nkt, int lane, TileFn&& tile,
                                         const f32x4 (&qcur)[sizeof(T) == 2 ? 4 : 8], f32x16 (&o)[2], float& mx, float& sum) {
    using Lay = AttnLayout<T>;
    const int half = lane >> 5, r32 = lane & 31;
    // ---- S^T = K . Q^T, tiles of 32 keys x 32 queries
    // (accumulators start from a literal zero C operand in the first MFMA of a tile: a separate zero-fill is 16 v_mov per
    // tile -- 176 issue slots per chunk in a kernel that is bound by vector issue, not by the matrix pipe)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NT];

    if constexpr (sizeof(T) == 2) {
        bf16x8 qf[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) qf[st] = __builtin_bit_cast(bf16x8, qcur[st]);
        // K fragments double-buffered: tile k+1's four ds_read_b128 are issued before tile k's MFMAs, so the LDS latency
        // hides under them (the plain read -> wait -> MFMA chain spent most of a chunk in s_waitcnt lgkmcnt)
        auto load_k = [&](int t, bf16x8 (&kf)[4]) {
            const char* kr = Ks + (t * 32 + r32) * Lay::kRowK;
            const int sw = k_swz(r32);                             // (32 t contributes nothing to the swizzle bits)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                // KPERM: the 64 d of a K row sit in accumulator order (qkv_attention_kernel writes them from its accumulators):
                // k-step st = (d tile st >> 1, register half st & 1) reads chunk 4 (st >> 1) + 2 half + (st & 1)
                const int ch = KPERM ? 4 * (st >> 1) + 2 * half + (st & 1) : 2 * st + half;
                kf[st] = *reinterpret_cast<const bf16x8*>(kr + ((ch ^ sw) << 4));
            }
        };
        bf16x8 kfa[4], kfb[4];
        if (tile(0) >= 0) load_k(tile(0), kfa);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int t = tile(k);
            const int tn = k + 1 < NT ? tile(k + 1) : -1;
            if (tn >= 0) { if (k & 1) load_k(tn, kfa); else load_k(tn, kfb); }
            __builtin_amdgcn_sched_barrier(0);  // the look-ahead reads go out BEFORE this tile's MFMAs
            if (t >= 0) {
                s[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((k & 1) ? kfb[0] : kfa[0], qf[0], zero16, 0, 0, 0);
#pragma unroll
                for (int st = 1; st < 4; ++st)
                    s[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((k & 1) ? kfb[st] : kfa[st], qf[st], s[k], 0, 0, 0);
            } else {
                s[k] = zero16;
            }
            __builtin_amdgcn_sched_barrier(0);  // one tile of look-ahead, no further hoisting (it would spill)
        }
    } else {
        // fp32: lane-half `half` owns d in [32*half, 32*half+32); MFMA m consumes d = 32*half + m
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int t = tile(k);
            s[k] = zero16;
            if (t >= 0) {
                const char* kr = Ks + (t * 32 + r32) * Lay::kRowK + (32 * half) * 4;
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + g * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        s[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qcur[g][e], s[k], 0, 0, 0);
                }
            }
        }
    }

    // ---- softmax numerators over these keys (registers x tiles in-lane, then the other lane-half).
    // exp2((s - max s) * log2(e)/8): the 1/sqrt(64) scale rides in the exp2 argument; only the last key tile of
    // the sequence can hold padded keys, so only it is masked.
    mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int t = tile(k);
        if (t >= 0) {
            if (t == nkt - 1) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = t * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                    if (key >= L) s[k][e] = -INFINITY;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[k][e]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    constexpr float kScaleLog2e = 0.125f * 1.4426950408889634f;
    const float mxs = mx * kScaleLog2e;
    sum = 0.f;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        if (tile(k) >= 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float p;
                // L = 256 + extras: the 9th key tile holds 1 or 2 real keys (registers 0, 1 of lane half 0); the other 14
                // registers are padded keys in both halves -- a literal 0 instead of exp2(-inf): a tenth of the kernel's
                // exponentials, which are what bounds it
                if (NKT == 9 && NT == 9 && k == 8 && e >= 2) {
                    s[k][e] = 0.f;
                    continue;
                }
                if constexpr (sizeof(T) == 2) p = __builtin_amdgcn_exp2f(fmaf(s[k][e], kScaleLog2e, -mxs));
                else p = expf((s[k][e] - mx) * 0.125f);
                s[k][e] = p;
                sum += p;
            }
        }
    }
    sum += __shfl_xor(sum, 32);

    // ---- O^T = V^T . P^T : two 32(d) x 32(q) tiles
    o[0] = zero16;
    o[1] = zero16;

    if constexpr (sizeof(T) == 2) {
        // V^T fragments double-buffered like the K fragments: [st][dt], keys t*32 + 16*st + 4*half + {0..3, 8..11}
        typedef __attribute__((ext_vector_type(4))) short s4;
        struct VF { s4 lo[2][2], hi[2][2]; };
        // lane = (half, dhalf, q, p): its 16-lane group reads the block keys 16 st + 4 half + {0..3} (hi: + 8) x d 16 dhalf .. + 15
        // of d-tile dt and supplies the address of row q, 8-byte piece p; lane i of the group receives column i, i.e. this
        // lane ends up with V[those 4 keys][dt * 32 + (lane & 31)] -- the V^T fragment the MFMA wants.
        const int tq = (lane >> 2) & 3, tp = lane & 3, dhalf = (lane >> 4) & 1;
        const char* vb0 = Vt + (4 * half + tq) * Lay::kRowV + 8 * (tp & 1);
        const int vsw = v_swz(4 * half + tq);                      // (the rows this lane addresses differ from 4 half + tq by multiples of 8)
        const char* vbd[2] = {vb0 + (((2 * dhalf + (tp >> 1)) ^ vsw) << 4), vb0 + (((4 + 2 * dhalf + (tp >> 1)) ^ vsw) << 4)};
        typedef __attribute__((address_space(3))) s4* lds_s4_ptr;
        auto load_v = [&](int t, VF& f) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* vr = vbd[dt] + (t * 32 + 16 * st) * Lay::kRowV;
                    f.lo[st][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(size_t)lds_addr_of(vr));
                    f.hi[st][dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(size_t)lds_addr_of(vr + 8 * Lay::kRowV));
                }
        };
        VF va, vb;
        if (tile(0) >= 0) load_v(tile(0), va);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int t = tile(k);
            const int tn = k + 1 < NT ? tile(k + 1) : -1;
            if (tn >= 0) { if (k & 1) load_v(tn, va); else load_v(tn, vb); }
            __builtin_amdgcn_sched_barrier(0);
            if (t >= 0) {
                const VF& f = (k & 1) ? vb : va;
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    // B operand: accumulator registers 8*st .. 8*st+7 as bf16; element j is key
                    // t*32 + 16*st + 8*(j>>2) + 4*half + (j&3)
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (short)f2bf(s[k][8 * st + j]);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        bf16x8 vf;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { vf[j] = f.lo[st][dt][j]; vf[4 + j] = f.hi[st][dt][j]; }
                        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int t = tile(k);
            if (t >= 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // registers 4g..4g+3 are keys t*32 + 8g + 4*half + (0..3): contiguous in V^T
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const char* vr = Vt + (dt * 32 + r32) * Lay::kRowV + (t * 32 + 8 * g + 4 * half) * 4;
                        const f32x4 vf = *reinterpret_cast<const f32x4*>(vr);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e], s[k][4 * g + e], o[dt], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// NKT > 0: the number of 32-key tiles is a compile-time constant (9 for L = 257/258: every shipped
// config), which removes the per-tile uniform guards, confines the padded-key mask to the last tile's
// 16 registers and takes ~1000 SGPR-spill lane moves out of the chunk body.  NKT == 0: generic L <= 288.
template <typename T, int NKT>
__global__ void __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1)
attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int B, int L, int H, int D, int Lp) {
    using Lay = AttnLayout<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;                                   // [kLP][kRowK]
    char* Vt = smem + kLP * Lay::kRowK;                // bf16: V [kLP][128 B] swizzled; fp32: V^T [64][kRowV]

    const int b = blockIdx.x / H, hh = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, r32 = lane & 31;
    constexpr long long ld = kHD;                                       // head-major: rows of one unit are contiguous
    const long long unit = (long long)Lp * kHD;                        // elements per (q | k | v, head) unit
    const T* qbase = qkv + ((long long)b * 3 * H + hh) * unit;
    const T* kbase = qbase + (long long)H * unit;
    const T* vbase = kbase + (long long)H * unit;
    const int nkt = NKT > 0 ? NKT : (L + 31) / 32;     // key tiles actually used
    constexpr int EPC = 16 / (int)sizeof(T);           // elements per 16-byte chunk
    constexpr int CPR = kHD / EPC;                     // chunks per row (8 bf16 / 16 fp32)

    // Q fragments of a 32-query chunk, straight from HBM/L2 (rows >= L are clamped and dropped later)
    constexpr int NQF = sizeof(T) == 2 ? 4 : 8;
    auto load_q = [&](int qc, f32x4 (&qf)[NQF]) {
        const int q = qc * 32 + r32;
        const T* qrow = qbase + (long long)(q < L ? q : L - 1) * ld;
#pragma unroll
        for (int i = 0; i < NQF; ++i)
            qf[i] = sizeof(T) == 2 ? *reinterpret_cast<const f32x4*>(qrow + 16 * i + 8 * half)
                                   : *reinterpret_cast<const f32x4*>(qrow + 32 * half + 4 * i);
    };
    f32x4 qnext[NQF];
    load_q(wave, qnext);   // first: in flight while the K/V tiles are being staged

    // ---- stage K (row-major) and V (transposed) of this head; zero the padded keys
    if constexpr (sizeof(T) == 2) {
        // LDS-DMA, 1 KB pieces of 8 rows x 128 B: piece p of K, then piece p of V; lane = (row 8p + (lane >> 3), slot lane & 7)
        // fetches the chunk that belongs in its slot (source-side swizzle).  Rows [L, Lp) of the images receive a copy of row L - 1 (masked keys);
        // rows [Lp, 32 nkt) of the LDS images are zeroed here (V: 0 x garbage must not be NaN).
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int np = Lp >> 3;                                        // pieces per operand (33 for L = 257 / 258)
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int lr = lane >> 3, slot = lane & 7;
        constexpr int MAXP = (kLP / 8 * 2 + 3) / 4;                    // pieces per wave, at most
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int p2 = wv + 4 * i;                                 // wave-uniform: [0, np) = K pieces, [np, 2 np) = V pieces
            if (p2 < 2 * np) {
                const bool isv = p2 >= np;
                const int p = isv ? p2 - np : p2, r = 8 * p + lr;
                const int ch = isv ? (slot ^ v_swz(r)) : (slot ^ k_swz(r));
                // (rows [L, Lp) of a unit are never written by the qkv Linear: those lanes fetch the last valid row instead -- finite values for keys
                // that the softmax masks; nothing may depend on what the workspace held before this call)
                const T* src = (isv ? vbase : kbase) + (long long)(r < L ? r : L - 1) * kHD + ch * 8;
                char* dst = (isv ? Vt : Ks) + p * 1024;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
            }
        }
        for (int i = tid; i < (nkt * 32 - Lp) * 16; i += 256) {       // 8 chunks per row, K and V
            const int rr = Lp + (i >> 4), c = i & 15;
            *reinterpret_cast<f32x4*>((c < 8 ? Ks : Vt) + rr * 128 + (c & 7) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's pieces have landed (hipcc does not track LDS-DMA writes)
    } else {
        for (int idx = tid; idx < nkt * 32 * CPR; idx += 256) {
            const int key = idx / CPR, ch = idx % CPR;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < L) {
                kv = *reinterpret_cast<const f32x4*>(kbase + (long long)key * ld + ch * EPC);
                vv = *reinterpret_cast<const f32x4*>(vbase + (long long)key * ld + ch * EPC);
            }
            *reinterpret_cast<f32x4*>(Ks + key * Lay::kRowK + ch * 16) = kv;
            T tmp[EPC];
            *reinterpret_cast<f32x4*>(tmp) = vv;
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                *reinterpret_cast<T*>(Vt + (ch * EPC + e) * Lay::kRowV + key * (int)sizeof(T)) = tmp[e];
        }
    }

    __syncthreads();

    // L = 32*8 + 1 or 2 (every shipped config): the 9th query chunk holds only the extra time / label tokens.  Given to
    // one wave it would cost a whole chunk (3 chunks on one wave against 2 on the others = +25 % on the workgroup); instead
    // all four waves take it together, each against its own key tiles, and the partial results are merged through LDS.
    constexpr bool SPLIT_LAST = NKT == 9;
    const int nqc = (L + 31) / 32;
    const int n_regular = SPLIT_LAST ? 8 : nqc;
    for (int qc = wave; qc < n_regular; qc += 4) {
        const int q = qc * 32 + r32;
        f32x4 qcur[NQF];
#pragma unroll
        for (int i = 0; i < NQF; ++i) qcur[i] = qnext[i];
        if (qc + 4 < n_regular) load_q(qc + 4, qnext);       // next chunk's Q lands under this chunk's work
        else if (SPLIT_LAST) load_q(8, qnext);

        f32x16 o[2];
        float mx, sum;
        attend_tiles<T, NKT, kMaxKeyTiles, false>(Ks, Vt, L, nkt, lane, [&](int k) { return k < nkt ? k : -1; }, qcur, o, mx, sum);
        const float inv = 1.0f / sum;

        // ---- store: lane = query, registers = d ; 4 consecutive d per register quad
        T* orow = out + ((long long)b * L + (q < L ? q : L - 1)) * D + hh * kHD;
        if constexpr (sizeof(T) == 2) {
            // register quads g, g + 1 hold d = 8g + 4 half .. and 8g + 8 + 4 half ..: v_permlane32_swap pairs the two lane halves
            // into 16 contiguous bytes per lane (d = 8g .. 8g+7 on half 0, 8g+8 .. 8g+15 on half 1), as the GEMM epilogue does
            // (all lanes take part in the swap; rows past L are dropped at the store)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                uint2 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16_t v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = f2bf(o[dt][4 * g + e] * inv);
                    v[g] = *reinterpret_cast<const uint2*>(v4);
                }
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(v[gp].x, v[gp + 1].x, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(v[gp].y, v[gp + 1].y, false, false);
                    const uint4 o4 = {s0[0], s1[0], s0[1], s1[1]};
                    if (q < L) *reinterpret_cast<uint4*>(orow + dt * 32 + 8 * gp + 8 * half) = o4;
                }
            }
        } else if (q < L) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = dt * 32 + 8 * g + 4 * half;
                    *reinterpret_cast<f32x4*>(orow + d) = f32x4{o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
                }
        }
    }

    if constexpr (SPLIT_LAST) {
        // wave w: key tiles w, w+4 and (wave 3) the 9th tile, which holds the keys of the extra tokens themselves
        float* part = reinterpret_cast<float*>(smem + kLP * Lay::kRowK + Lay::kVBytes);   // [4 waves][2 queries][64 d + max + sum]
        const int rem = L - 256;                                                               // 1 or 2 valid queries
        f32x16 o[2];
        float mx, sum;
        attend_tiles<T, NKT, 3, false>(Ks, Vt, L, nkt, lane, [&](int k) { return k < 2 ? wave + 4 * k : (wave == 3 ? 8 : -1); }, qnext, o, mx, sum);
        if (r32 < rem) {
            float* pw = part + (wave * 2 + r32) * 66;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(pw + dt * 32 + 8 * g + 4 * half) =
                        f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
            if (half == 0) { pw[64] = mx; pw[65] = sum; }
        }
        __syncthreads();
        if (tid < rem * 64) {
            const int qi = tid >> 6, d = tid & 63;
            float m[4], M = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; ++w) { m[w] = part[(w * 2 + qi) * 66 + 64]; M = fmaxf(M, m[w]); }
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float f;
                if constexpr (sizeof(T) == 2) f = __builtin_amdgcn_exp2f((m[w] - M) * (0.125f * 1.4426950408889634f));
                else f = expf((m[w] - M) * 0.125f);
                num = fmaf(f, part[(w * 2 + qi) * 66 + d], num);
                den = fmaf(f, part[(w * 2 + qi) * 66 + 65], den);
            }
            out[((long long)b * L + 256 + qi) * D + hh * kHD + d] = Elem<T>::from_f32(num / den);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// attn.qkv + attention in one launch (bf16, L = 256 patches + 1 or 2 extra tokens, head_dim 64).
//
// reference models/uvit.py:152-164: qkv = Linear(norm1(x)); q, k, v = split; softmax(q k^T / 8) v.  The plain path writes
// the [B L, 3 D] qkv tensor (101 MB at B = 128) and reads it back one launch later; here a workgroup = one (image, head)
// computes its own 256 x 192 slice of the Linear from the norm1 rows and keeps it on the CU:
//   phase A  8 waves x 32 patch rows.  The wave's rows of h = norm1(x) sit in registers as MFMA B fragments (32 k-steps);
//            the head's six 32-column weight tiles (q0 q1 k0 k1 v0 v1; host-packed in fragment order, qkv_attention_pack)
//            stream through a 2 x 32 KB LDS ring by LDS-DMA, every wave multiplies each tile against its rows:
//            out^T[col, row] so that a lane owns a row.  q stays in registers (already the B fragments of S^T = K Q^T, in
//            accumulator k order), k and v are written to the LDS images the attention core reads (K rows in that same
//            accumulator order, V row-major).
//   extras   the 1 or 2 extra tokens' rows of h (row-major, written by the small reduce / skip_rows launches) are parked in LDS; their
//            q / k / v are computed here too, as a ninth 32-row group whose k range is split over the 8 waves (4 extra MFMAs per
//            tile and wave; partial sums through LDS): their K / V rows go behind the patch rows in the images (softmax is
//            order-free), their queries are the split last chunk.
//   phase B  attend_tiles, one 32-query chunk per wave, then the split chunk of the extra tokens over all 8 waves.
// Image rows: r in [0, 256) = token E + r, rows 256 .. 256 + E - 1 = tokens 0 .. E - 1, the rest zero (masked).
struct QkvAttnArgs {
    const bf16_t* h;       // norm1 of the patch rows in fragment order: [B * 8 groups of 32 rows][D / 16][64 lanes][8] (MlpFusedArgs::ln_out_frag)
    const bf16_t* wimg;    // [H][6 tiles][D / 16 k-steps][64 lanes][8] (qkv_attention_pack)
    const float* bias;     // [3 D] or nullptr
    const bf16_t* hx;      // norm1 rows, row-major [B L, D]: only the extra-token rows (l < E) are read -- or nullptr:
    const float* xres;     //   then the kernel normalises those rows itself from the fp32 residual stream [B L, D]
    const float* ln_g;     //   with this block's norm1 weight
    const float* ln_b;     //   and bias (reference models/uvit.py:206, eps 1e-5, two-pass statistics in fp32)
    bf16_t* out;           // [B L, D]
    int B, L, H, Lp, E;
};

// LDS behind the K / V images: px = the extra rows' partial q / k / v sums [8 waves][6 tiles][2 rows][32 columns] fp32 (the split
// chunk's [8 waves][2 queries][64 d, max, sum] fp32 reuses its start later) | hxl = the extra rows of h [2][D] bf16 | qxl = their q [2][64] bf16
constexpr int kQaPxBytes = 8 * 6 * 2 * 32 * 4;
constexpr int kQaHxPad = 64;                         // the second extra row starts 16 banks off the first: a lane group of the 16x16x32 B-operand read holds both rows at two 16-byte offsets
constexpr int kQaHxBytes = 2 * (1024 * 2 + kQaHxPad);  // two rows of D <= 1024 bf16
constexpr int kQaAuxBytes = kQaPxBytes + kQaHxBytes + 256;
constexpr int kQaKQ = 8;                             // k-steps per slice of the k range
constexpr int kQaBlk = 2 * kQaKQ * 1024;             // one weight block = (slice, tile pair): 2 x 8 k-steps x 1 KB fragments
constexpr int kQaRing = 4 * kQaBlk;
static_assert(8 * 2 * 66 * 4 <= kQaPxBytes, "the split chunk's partials reuse the px area");

typedef const __attribute__((address_space(1))) void* qa_gptr_t;
typedef __attribute__((address_space(3))) void* qa_lptr_t;

// Phase A since round 4 -- the k range in SLICES of 8 k-steps, all six accumulators resident, the rows double-buffered:
//   for slice q: [rows of h, k-steps 8 q .. 8 q + 7, as 32 VGPRs of B fragments; slice q + 1 is requested into the other 32 now]
//       for tile pair (q0 q1), (k0 k1), (v0 v1):  acc[t] += W(q, t) . h_q^T      (one 16 KB weight block = 2 tiles x 8 k-steps, 16 MFMAs per wave)
// The row fetch is bandwidth, not latency (256 KB per workgroup at the ~23 B/clk a CU takes in: 5 us, which the round-3 kernel waited
// out before its first MFMA): here the first MFMA waits for an eighth of it (D = 512: a quarter) and every later slice arrives under
// the three blocks of the slice before.  The weight blocks stream through a ring of four 16 KB slots requested three blocks ahead
// (counted vmcnt; the old two-slot ring of whole 32 KB tiles was requested one tile ahead and drained with vmcnt(0)), and the bias /
// pack / LDS-image epilogue runs once, after the last slice.  It is also what lets embed_dim 768 / 1024 in: their rows (192 / 256
// fragment registers) never fit beside the accumulators at two waves per SIMD.
// One barrier per block, in its MIDDLE (behind the 8th of its 16 MFMAs): by then every wave has finished block b - 1, so the LDS-DMA of
// block b + 3 may overwrite that slot, and block b + 1 is confirmed landed half a block before its first fragment is read -- the
// fragment queue (four ds_read_b128 ahead of their MFMAs) runs on across the block boundaries without a bubble.
// VMEM bookkeeping (hipcc counts neither the asm loads nor what an LDS-DMA piece covers): every vmcnt below is the exact number of
// requests issued behind the one waited for, from the fixed issue order of a block:
//   [first block of a slice: wait: this slice's rows; request the next slice's (8 loads)] 8 MFMAs [wait: LDS-DMA of block + 1] barrier
//   [LDS-DMA of block + 3 (2 pieces)] [extras' MFMAs] 8 MFMAs
constexpr int qa_dma_ops(int b, int nb) { return b < nb ? 2 : 0; }                         // LDS-DMA pieces per wave of block b (none past the end)
constexpr int qa_pref_ops(int b, int nb) { return (b >= 0 && b % 3 == 0 && b / 3 + 1 < nb / 3) ? kQaKQ : 0; }   // row requests issued at the start of block b (for the next slice)
// requests younger than the LDS-DMA of block b + 1 (issued in the middle of block b - 2) when the middle of block b waits for it
constexpr int qa_younger_dma(int b, int nb) { return qa_pref_ops(b - 1, nb) + qa_dma_ops(b + 2, nb) + qa_pref_ops(b, nb); }
// requests younger than the rows of the slice that starts at block b (requested at the start of block b - 3) when block b waits for them
constexpr int qa_younger_rows(int b, int nb) { return qa_dma_ops(b, nb) + qa_dma_ops(b + 1, nb) + qa_dma_ops(b + 2, nb); }

template <int D>
__global__ void __launch_bounds__(512, 1) qkv_attention_kernel(const QkvAttnArgs a) {
    using Lay = AttnLayout<bf16_t>;
    constexpr int KS = D / 16, NS = KS / kQaKQ, NB = 3 * NS;   // k-steps, slices, weight blocks of this head
    static_assert(D % 128 == 0 && D >= 512 && D <= 1024, "k range in slices of 8 k-steps; hxl holds two rows of D <= 1024");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vt = smem + kLP * Lay::kRowK;
    float* px = reinterpret_cast<float*>(smem + kLP * Lay::kRowK + Lay::kVBytes);
    float* part = px;
    char* hxl = smem + kLP * Lay::kRowK + Lay::kVBytes + kQaPxBytes;
    char* qxl = hxl + kQaHxBytes;
    char* ring = smem + kLP * Lay::kRowK + Lay::kVBytes + kQaAuxBytes;

    // XCD-aware placement (workgroup i runs on XCD i % 8): the H heads of an image read the same rows of h, so they take
    // consecutive slots of ONE XCD and its L2 fetches those rows once
    int b, hh;
    if ((a.B & 7) == 0) { const int slot = blockIdx.x >> 3; b = (blockIdx.x & 7) + 8 * (slot / a.H); hh = slot % a.H; }
    else { b = blockIdx.x / a.H; hh = blockIdx.x % a.H; }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, r32 = lane & 31;
    const int L = a.L, E = a.E;

    // LDS-DMA in the scalar-base form (uniform 64-bit base in SGPRs + one 32-bit lane offset), written as asm: the builtin makes a 64-bit VGPR address
    // pair of it, and that form serialises with the MFMAs of BOTH waves of the SIMD (75 cycles per request, profiles/r05/dma_mfma_probe_roles.txt)
    const char* wsrc = reinterpret_cast<const char*>(a.wimg) + (size_t)hh * NB * kQaBlk;
    const unsigned lane16 = lane * 16;
    auto dma_block = [&](int bb) {      // block bb -> ring slot bb & 3: two 1 KB pieces per wave
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pc = wave * 2 + i;
            const char* sb = wsrc + (size_t)bb * kQaBlk + pc * 1024;
            const unsigned lds = lds_addr_of(ring + (bb & 3) * kQaBlk + pc * 1024);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(lane16), "s"(sb) : "memory", "m0");
        }
    };
    dma_block(0);
    dma_block(1);
    dma_block(2);

    // ---- this wave's 32 rows of h, part 0, as B fragments (k-step ks: k = 16 ks + 8 half .. + 7): the producer left them in exactly
    // this order ([32-row group][k-step][lane] x 16 bytes: MlpFusedArgs::ln_out_frag / launch_layernorm_frag), 1 KB per load instruction
    const int row = 32 * wave + r32;                                   // image row = patch index
    const bf16x8* hfr_u = reinterpret_cast<const bf16x8*>(a.h) + ((long long)b * 8 + wave) * KS * 64;     // wave-uniform: the lane's 16 bytes ride in the 32-bit offset
    const bf16x8* hfr = hfr_u + lane;
    bf16x8 xs[2][kQaKQ];                                               // slice q lives in set q & 1
#pragma unroll
    for (int ks = 0; ks < kQaKQ; ++ks) xs[0][ks] = hfr[ks * 64];

    // ---- the extra tokens' rows of h into LDS (every wave reads its k range of them in every block); zero rows behind the images' last
    if (a.hx) {
        for (int i = tid; i < E * (D / 8); i += 512) {
            const int e = i / (D / 8), cch = i % (D / 8);
            *reinterpret_cast<f32x4*>(hxl + e * (D * 2 + kQaHxPad) + cch * 16) = *reinterpret_cast<const f32x4*>(a.hx + ((long long)b * L + e) * D + cch * 8);
        }
    } else if (wave < E) {     // wave e normalises extra row e (two-pass statistics, fp32): D / 64 columns per lane
        constexpr int NV = D / 256;
        const float* xr = a.xres + ((long long)b * L + wave) * D + lane * (D / 64);
        f32x4 xv[NV];
        float s1 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            xv[v] = *reinterpret_cast<const f32x4*>(xr + 4 * v);
            s1 += (xv[v][0] + xv[v][1]) + (xv[v][2] + xv[v][3]);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o);
        const float mean = s1 / (float)D;
        float s2 = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            xv[v] = xv[v] - mean;
            s2 += (xv[v][0] * xv[v][0] + xv[v][1] * xv[v][1]) + (xv[v][2] * xv[v][2] + xv[v][3] * xv[v][3]);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
        const float rstd = 1.0f / sqrtf(s2 / (float)D + 1e-5f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(a.ln_g + lane * (D / 64) + 4 * v), c0 = *reinterpret_cast<const f32x4*>(a.ln_b + lane * (D / 64) + 4 * v);
            const f32x4 y = xv[v] * rstd * g + c0;
            *reinterpret_cast<uint2*>(hxl + wave * (D * 2 + kQaHxPad) + (lane * (D / 64) + 4 * v) * 2) = uint2{pack2_bf16(y[0], y[1]), pack2_bf16(y[2], y[3])};
        }
    }
    for (int i = tid; i < (kLP - 256 - E) * 16; i += 512) {
        const int r = 256 + E + (i >> 4), c = i & 15;
        *reinterpret_cast<f32x4*>((c < 8 ? Ks : Vt) + r * 128 + (c & 7) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- phase A
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = zero16;
    // extras on 16x16x32 MFMAs: lane l supplies A = W[column 16 (wave & 1) + (l & 15)][k = 32 (wave >> 1) + 8 (l >> 4) .. + 7] of a tile, which sits
    // in fragment (k-step 2 (wave >> 1) + (l >> 5)) at lane slot (column + 32 ((l >> 4) & 1)); and B = h[extra row (l & 15)][the same k]
    f32x4 accx[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) accx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int xg = lane >> 4, xn = lane & 15;
    const int wx16 = (2 * (wave >> 1) + (xg >> 1)) * 1024 + ((16 * (wave & 1) + xn) + 32 * (xg & 1)) * 16;
    const char* hx16 = hxl + (xn < E ? xn : 0) * (D * 2 + kQaHxPad) + (wave >> 1) * 64 + xg * 16;
    bf16x8 wq[4];            // the fragment queue: fragment g of the stream (block g >> 4, fragment g & 15) is read four MFMAs ahead
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // blocks 0..2, slice 0 of the rows, hxl, the zero rows
#pragma unroll
    for (int i = 0; i < 4; ++i) wq[i] = *reinterpret_cast<const bf16x8*>(ring + lane * 16 + i * 1024);
    [&]<int... BI>(std::integer_sequence<int, BI...>) {
        ([&] {
            constexpr int bb = BI, q = bb / 3, jj = bb % 3, S = q & 1;
            if constexpr (jj == 0) {
                if constexpr (q > 0) {      // this slice's rows (requested three blocks ago): one wait, tied to the registers it guards
                    asm volatile("s_waitcnt vmcnt(%8)"
                                 : "+v"(xs[S][0]), "+v"(xs[S][1]), "+v"(xs[S][2]), "+v"(xs[S][3]), "+v"(xs[S][4]), "+v"(xs[S][5]), "+v"(xs[S][6]), "+v"(xs[S][7])
                                 : "i"(qa_younger_rows(bb, NB)));
                }
                if constexpr (q + 1 < NS) {  // the next slice's rows into the other set (last read in the block before this one)
#pragma unroll
                    for (int ks = 0; ks < kQaKQ; ++ks)
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xs[S ^ 1][ks]) : "v"(lane16), "s"(hfr_u + ((q + 1) * kQaKQ + ks) * 64) : "memory");
                }
            }
            const char* wb = ring + (bb & 3) * kQaBlk + lane * 16;
            const char* wbn = ring + ((bb + 1) & 3) * kQaBlk + lane * 16;
#pragma unroll
            for (int f = 0; f < 2 * kQaKQ; ++f) {
                if (f == kQaKQ) {
                    if constexpr (bb + 1 < NB) {
                        // this wave's pieces of block bb + 1 have landed (bb + 1 < 3: with the wait in front of block 0) | everyone's have, and
                        // everyone is past block bb - 1
                        if constexpr (bb + 1 >= 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(qa_younger_dma(bb, NB)) : "memory");
                        else asm volatile("s_barrier" ::: "memory");
                    }
                    if constexpr (bb + 3 < NB) dma_block(bb + 3);
                    // the extra tokens' rows against this block, on v_mfma_f32_16x16x32_bf16 (M = 16 of the tile's 32 columns, N = 16 "rows" of
                    // which the first E are the extra tokens, K = 32): wave w takes column half w & 1 and k-block w >> 1 of the slice's four, one
                    // small MFMA per tile, accumulated over ALL slices in 4 registers per tile -- no LDS round trip per block (the 32x32 form
                    // with its partial sums read-modify-written in LDS cost 45 % of phase A for 0.4 % of the rows); the waves' partial sums meet
                    // in LDS once, after the last slice.  The A fragment is gathered from the block's 32x32x16 fragment order, 16 bytes per lane.
                    const bf16x8 hb = *reinterpret_cast<const bf16x8*>(hx16 + q * (kQaKQ * 32));
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(ring + (bb & 3) * kQaBlk + t * (kQaKQ * 1024) + wx16);
                        accx[2 * jj + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, hb, accx[2 * jj + t], 0, 0, 0);
                    }
                }
                const bf16x8 w0 = wq[f & 3];
                if (f + 4 < 2 * kQaKQ) wq[f & 3] = *reinterpret_cast<const bf16x8*>(wb + (f + 4) * 1024);
                else if (bb + 1 < NB) wq[f & 3] = *reinterpret_cast<const bf16x8*>(wbn + (f + 4 - 2 * kQaKQ) * 1024);   // the next block's first fragments
                // (no sched_barrier around the MFMA: hipcc's own placement of the fragment reads measures 3.5 % faster on a full grid)
                acc[2 * jj + (f >> 3)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xs[S][f & 7], acc[2 * jj + (f >> 3)], 0, 0, 0);
            }
        }(), ...);
    }(std::make_integer_sequence<int, NB>{});

    // the extras' partial sums of this wave: lanes (l & 15) < E hold, for tile j, columns 16 (wave & 1) + 4 (l >> 4) .. + 3 of extra row l & 15
    if (xn < E) {
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(px + (((wave * 6 + j) * 2 + xn) * 16 + 4 * xg)) = accx[j];
    }
    // ---- q stays in registers (its accumulators, packed to bf16, are the B fragments of S^T = K Q^T in accumulator k order), k and v
    // go to the LDS images the attention core reads (K rows in that same accumulator order, V row-major)
    f32x4 qcur[4];
    const int swk = k_swz(row), swv = v_swz(row);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        if (a.bias) {
            const float* bj = a.bias + (j >> 1) * D + hh * kHD + 32 * (j & 1) + 4 * half;   // register e: column (e & 3) + 8 (e >> 2) + 4 half
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bj + 8 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][4 * g + e] += bv[e];
            }
        }
        unsigned pk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) pk[i] = pack2_bf16(acc[j][2 * i], acc[j][2 * i + 1]);
        const int T = j & 1;
        if (j < 2) {            // q: registers 8 eh .. 8 eh + 7 of d tile T are k-step 2 T + eh of S^T = K Q^T
            qcur[2 * T] = __builtin_bit_cast(f32x4, uint4{pk[0], pk[1], pk[2], pk[3]});
            qcur[2 * T + 1] = __builtin_bit_cast(f32x4, uint4{pk[4], pk[5], pk[6], pk[7]});
        } else if (j < 4) {     // k: the same order, chunk 4 T + 2 half + eh of the row
            *reinterpret_cast<uint4*>(Ks + row * 128 + (((4 * T + 2 * half) ^ swk) << 4)) = uint4{pk[0], pk[1], pk[2], pk[3]};
            *reinterpret_cast<uint4*>(Ks + row * 128 + (((4 * T + 2 * half + 1) ^ swk) << 4)) = uint4{pk[4], pk[5], pk[6], pk[7]};
        } else {                // v: row-major, register quad g = d 32 T + 8 g + 4 half .. + 3; v_permlane32_swap pairs the lane halves into whole 16-byte
                                // chunks (lane half 0: chunk 4 T + gp, half 1: chunk 4 T + gp + 1), so that 8 consecutive rows hit 8 different chunks
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * gp], pk[2 * gp + 2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * gp + 1], pk[2 * gp + 3], false, false);
                *reinterpret_cast<uint4*>(Vt + row * 128 + (((4 * T + gp + half) ^ swv) << 4)) = uint4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }
    __syncthreads();     // every wave's partial sums of the extra rows are in px
    if (tid < E * 192) {
        const int e = tid / 192, c = tid % 192, j = c >> 5, col = c & 31;
        float v = a.bias ? a.bias[(j >> 1) * D + hh * kHD + 32 * (j & 1) + col] : 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) v += px[(((2 * kb + (col >> 4)) * 6 + j) * 2 + e) * 16 + (col & 15)];   // k-blocks in ascending order
        const bf16_t vb = f2bf(v);
        const int d = 32 * (j & 1) + col, r = 256 + e;
        if (j < 2) {
            *reinterpret_cast<bf16_t*>(qxl + e * 128 + d * 2) = vb;
        } else if (j < 4) {      // K row in accumulator order (KPERM): d = 32 T + (ae & 3) + 8 (ae >> 2) + 4 hf -> chunk 4 T + 2 hf + (ae >> 3), element ae & 7
            const int T = d >> 5, dd = d & 31, hf = (dd >> 2) & 1, ae = (dd & 3) + 4 * (dd >> 3);
            *reinterpret_cast<bf16_t*>(Ks + r * 128 + (((4 * T + 2 * hf + (ae >> 3)) ^ k_swz(r)) << 4) + (ae & 7) * 2) = vb;
        } else {
            *reinterpret_cast<bf16_t*>(Vt + r * 128 + (((d >> 3) ^ v_swz(r)) << 4) + (d & 7) * 2) = vb;
        }
    }
    __syncthreads();     // the K / V images are complete

    // the extra tokens' queries (the split chunk at the end), in accumulator k order like the patch rows' own
    f32x4 qe[4];
    {
        const bf16_t* qrow = reinterpret_cast<const bf16_t*>(qxl) + (r32 < E ? r32 : E - 1) * kHD + 4 * half;
#pragma unroll
        for (int st = 0; st < 4; ++st) {     // k-step st = (T = st >> 1, eh = st & 1): d 32 T + 16 eh + 4 half + {0..3} and + 8
            const uint2 lo = *reinterpret_cast<const uint2*>(qrow + 32 * (st >> 1) + 16 * (st & 1));
            const uint2 hi = *reinterpret_cast<const uint2*>(qrow + 32 * (st >> 1) + 16 * (st & 1) + 8);
            qe[st] = __builtin_bit_cast(f32x4, uint4{lo.x, lo.y, hi.x, hi.y});
        }
    }

    // ---- phase B: this wave's 32 patch queries against all 9 key tiles
    {
        f32x16 o[2];
        float mx, sum;
        attend_tiles<bf16_t, 9, kMaxKeyTiles, true>(Ks, Vt, L, kMaxKeyTiles, lane, [&](int k) { return k; }, qcur, o, mx, sum);
        const float inv = 1.0f / sum;
        bf16_t* orow = a.out + ((long long)b * L + E + row) * D + hh * kHD;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            uint2 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16_t v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = f2bf(o[dt][4 * g + e] * inv);
                v[g] = *reinterpret_cast<const uint2*>(v4);
            }
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(v[gp].x, v[gp + 1].x, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(v[gp].y, v[gp + 1].y, false, false);
                *reinterpret_cast<uint4*>(orow + dt * 32 + 8 * gp + 8 * half) = uint4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }

    // ---- the extra tokens' queries: all 8 waves together, wave w against key tile w (wave 7: and the 9th), merged through LDS
    {
        f32x16 o[2];
        float mx, sum;
        attend_tiles<bf16_t, 9, 2, true>(Ks, Vt, L, kMaxKeyTiles, lane, [&](int k) { return k == 0 ? wave : (wave == 7 ? 8 : -1); }, qe, o, mx, sum);
        if (r32 < E) {
            float* pw = part + (wave * 2 + r32) * 66;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(pw + dt * 32 + 8 * g + 4 * half) =
                        f32x4{o[dt][4 * g], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]};
            if (half == 0) { pw[64] = mx; pw[65] = sum; }
        }
        __syncthreads();
        if (tid < E * 64) {
            const int qi = tid >> 6, d = tid & 63;
            float m[8], M = -INFINITY;
#pragma unroll
            for (int w = 0; w < 8; ++w) { m[w] = part[(w * 2 + qi) * 66 + 64]; M = fmaxf(M, m[w]); }
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float f = __builtin_amdgcn_exp2f((m[w] - M) * (0.125f * 1.4426950408889634f));
                num = fmaf(f, part[(w * 2 + qi) * 66 + d], num);
                den = fmaf(f, part[(w * 2 + qi) * 66 + 65], den);
            }
            a.out[((long long)b * L + qi) * D + hh * kHD + d] = f2bf(num / den);
        }
    }
}

}  // namespace

template <typename T>
hipError_t launch_attention(const T* qkv, T* out, int B, int L, int H, int D, hipStream_t s) {
    if (L > kLP || D != H * kHD || L < 1) return hipErrorInvalidValue;
    using Lay = AttnLayout<T>;
    const size_t lds = (size_t)kLP * Lay::kRowK + (size_t)Lay::kVBytes + kPartBytes;
    const int Lp = make_head_major(L, H).Lp;
    // the <T, 9> specialisation assumes L = 256 + 1 or 2 (one or two real keys / queries in the 9th tile): every shipped config
    if (L == 257 || L == 258) hipLaunchKernelGGL((attention_kernel<T, 9>), dim3(B * H), dim3(256), lds, s, qkv, out, B, L, H, D, Lp);
    else hipLaunchKernelGGL((attention_kernel<T, 0>), dim3(B * H), dim3(256), lds, s, qkv, out, B, L, H, D, Lp);
    return hipGetLastError();
}

// attn.qkv weight [3 D, D] (nn.Linear layout) -> per head the weight blocks the kernel streams, in stream order: [slice q of the k range:
// 8 k-steps][tile pair jj][tile t of the pair][k-step][lane] x 16 bytes (tiles j = 2 jj + t = q0 q1 k0 k1 v0 v1), each 1 KB fragment in
// MFMA A-operand order:
// img[(((((head * NS + q) * 3 + jj) * 2 + t) * 8 + ks) * 64 + lane) * 8 + i] = W[(j >> 1) D + 64 head + 32 (j & 1) + (lane & 31)][16 (8 q + ks) + 8 (lane >> 5) + i]
void qkv_attention_pack(int D, int H, const float* w, unsigned short (*to_bf16)(float), unsigned short* img) {
    const int NS = D / 16 / kQaKQ;
    for (int hh = 0; hh < H; ++hh)
        for (int q = 0; q < NS; ++q)
            for (int j = 0; j < 6; ++j)
                for (int ks = 0; ks < kQaKQ; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const float* src = w + ((size_t)(j >> 1) * D + 64 * hh + 32 * (j & 1) + (lane & 31)) * D + 16 * (q * kQaKQ + ks) + 8 * (lane >> 5);
                        unsigned short* dst = img + (((((size_t)hh * NS + q) * 6 + j) * kQaKQ + ks) * 64 + lane) * 8;
                        for (int i = 0; i < 8; ++i) dst[i] = to_bf16(src[i]);
                    }
}

bool qkv_attention_supported(int D, int H, int L, int extras) {
    return (D == 512 || D == 768 || D == 1024) && H * kHD == D && (extras == 1 || extras == 2) && L == 256 + extras;
}

static size_t qkv_attention_lds() { return (size_t)kLP * AttnLayout<bf16_t>::kRowK + AttnLayout<bf16_t>::kVBytes + kQaAuxBytes + kQaRing; }

hipError_t launch_qkv_attention(const bf16_t* h, const bf16_t* wimg, const float* bias, const bf16_t* hx, const float* xres,
                                const float* ln_g, const float* ln_b, bf16_t* out, int B, int L, int H, int D, int extras, hipStream_t s) {
    if (!qkv_attention_supported(D, H, L, extras) || !h || !wimg || (!hx && (!xres || !ln_g || !ln_b)) || !out || B < 1) return hipErrorInvalidValue;
    const QkvAttnArgs a{h, wimg, bias, hx, xres, ln_g, ln_b, out, B, L, H, make_head_major(L, H).Lp, extras};
    switch (D) {
        case 512: hipLaunchKernelGGL((qkv_attention_kernel<512>), dim3(B * H), dim3(512), qkv_attention_lds(), s, a); break;
        case 768: hipLaunchKernelGGL((qkv_attention_kernel<768>), dim3(B * H), dim3(512), qkv_attention_lds(), s, a); break;
        default: hipLaunchKernelGGL((qkv_attention_kernel<1024>), dim3(B * H), dim3(512), qkv_attention_lds(), s, a); break;
    }
    return hipGetLastError();
}

hipError_t init_attention_kernels() {
    for (const void* f : {(const void*)qkv_attention_kernel<512>, (const void*)qkv_attention_kernel<768>, (const void*)qkv_attention_kernel<1024>})
        if (hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)qkv_attention_lds()); e != hipSuccess) return e;
    const int lb = kLP * AttnLayout<bf16_t>::kRowK + AttnLayout<bf16_t>::kVBytes + kPartBytes;
    const int lf = kLP * AttnLayout<float>::kRowK + AttnLayout<float>::kVBytes + kPartBytes;
    hipError_t e = hipFuncSetAttribute((const void*)attention_kernel<bf16_t, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, lb);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attention_kernel<bf16_t, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lb);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attention_kernel<float, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, lf);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attention_kernel<float, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lf);
    return e;
}

template hipError_t launch_attention<bf16_t>(const bf16_t*, bf16_t*, int, int, int, int, hipStream_t);
template hipError_t launch_attention<float>(const float*, float*, int, int, int, int, hipStream_t);

}  // namespace dd
