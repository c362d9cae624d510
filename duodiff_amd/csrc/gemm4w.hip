// Persistent bf16 MFMA GEMM for the wide U-ViT Linears (embed_dim 768 / 1024: ImageNet-64, ImageNet-256 latents) on gfx950:
//     C[M, N] = [A | A2][M, K] . W[N, K]^T  (+ the fused epilogues of gemm.hip: bias / exact-erf GELU / residual add / head-major qkv)
// reference models/uvit.py:86-92 (fc1, fc2), :155-168 (attn.qkv, attn.proj), :196-200 (skip_linear).
//
// Why a second GEMM kernel.  gemm256 (gemm.hip) double-buffers 64 KB k-tiles in LDS with LDS-DMA: a k-tile is requested ONE k-tile
// time (~1 us of MFMA work) before it is needed and every wave then waits on vmcnt(0) + a barrier.  On the D = 768 / 1024 shapes
// the A operand streams from HBM (the 406 MB hidden activation of fc2, the LayerNorm output of qkv / fc1), its latency under load
// is 2-3 us, and the kernel runs at 2-2.8 us per k-tile whatever the MFMA pipe could do (profiles/r04: 0.31-0.35 of the bf16 roof
// even at K = 3072, where the epilogue is 3 % of a tile).  160 KB of LDS cannot hold a third 64 KB stage, so the extra depth
// comes from REGISTERS here:
//   * 4 waves, one per SIMD (the 512-entry register file each), wave tile 128 x 32 TN: the accumulators are AGPR-resident
//     (192 for TN = 3, + 32 for the tail rows), which leaves the 256 VGPRs for staging;
//   * every operand byte travels global -> VGPR -> LDS: k-tile u+3 is requested (global_load_dwordx4, 16 B / lane, the same
//     8-rows-x-128-B access shape and source-side XOR swizzle as gemm256's LDS-DMA) while k-tile u is multiplied, sits in one of two
//     register sets for two k-tile times, and is written (ds_write_b128, conflict-free: a wave instruction = 1 KB linear) into
//     the LDS stage k-tile u-1 has left while k-tile u+... runs: load -> use distance = 2 k-tiles of MFMA work instead of 1;
//   * the k-tile body is straight-line code, one asm volatile statement per instruction, in the order written: 4 k-steps x
//     (4 TN + tail) MFMAs, behind each of them at most one ds_read_b128 (the fragments of the NEXT k-step, so only counted
//     lgkmcnt waits remain) or one staging slot (s_waitcnt vmcnt(2 NL - 1) ; ds_write_b128 ; global_load_dwordx4 into the register
//     just written out).  ONE barrier per k-tile, at the start of its last k-step: by then every wave has written its share of
//     the next stage and holds the last fragments of this one in registers, so the last k-step prefetches the next k-tile's first
//     fragments from the other stage and no LDS latency is exposed at the k-tile boundary;
//   * persistent workgroups walk their tiles with the k-tile stream continuous across tile boundaries (the next tile's first
//     three k-tiles are in flight during the epilogue), XCD-aware tile order, exact row partition with tail rows (gemm256's).
// Tile 256 x 64 TN, TN = 3 (N % 192 == 0: 768, 2304, 3072) or 2 (N % 128 == 0: 1024, 4096): with the tail-row partition every
// shipped shape gets a tile count that is a multiple of 256 CUs (ImageNet-64: 256 x {12, 16, 4} tiles; ImageNet-256 latents at
// B = 32: 32 x {16 (qkv), 32 (fc1), 8 (proj / fc2 / skip)} -- gemm256 ran the N = 1024 Linears on 128 of the 256 CUs).
#include "dd_internal.h"

#include <type_traits>
#include <utility>

namespace dd {
namespace {

constexpr int kARows = 264;                 // 256 main rows + 8 tail rows
constexpr int kABytes = kARows * 128;       // one A stage (BK = 64 bf16 = 128 B per row)

template <int TN>
struct G4 {
    static constexpr int BN = 64 * TN;
    static constexpr int WBytes = BN * 128;
    // LDS: [A stage 0][A stage 1][W stage 0][W stage 1] -- the two stages of an operand are adjacent so that a stage is an immediate
    // offset (< 64 KB) from one per-lane base register
    static constexpr int WBase = 2 * kABytes;
    static constexpr int Lds = 2 * kABytes + 2 * WBytes;
};

__device__ __forceinline__ unsigned lds_addr4(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ void mfma4(f32x16& acc, const bf16x8& w, const bf16x8& x) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
}
template <int N>
__device__ __forceinline__ void wait_lgkm4() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm4() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }
template <int OFF>
__device__ __forceinline__ void lds_rd4(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
template <int OFF>
__device__ __forceinline__ void lds_wr4(unsigned addr, const f32x4& d) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(d), "i"(OFF) : "memory");
}
// 16 bytes per lane from (uniform base in SGPRs) + (32-bit lane offset); hipcc does not count asm loads: the slots wait themselves
__device__ __forceinline__ void gload4(f32x4& d, unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase));
}

__device__ __forceinline__ uint2 pack4_bf16(const f32x4& q) {
    typedef __bf16 bf16v4 __attribute__((ext_vector_type(4)));
    const bf16v4 b = __builtin_convertvector(q, bf16v4);
    return __builtin_bit_cast(uint2, b);
}

// exact-erf GELU for a bf16 result: the polynomial of gemm.hip's epilogue (|gelu error| <= 2.4e-4)
__device__ __forceinline__ f32x4 gelu4(f32x4 v) {
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

template <int TN, int EPI, bool TAIL>
__global__ void __launch_bounds__(256) gemm4w_kernel(const GemmArgs<bf16_t> a, const int q_tiles, const int e_tail, const int tail_base) {
    using C = G4<TN>;
    constexpr int NX = TAIL ? (TN + 1) / 2 : 0;          // tail-row MFMAs per k-step and wave (the 2 TN tail column tiles of a workgroup over 4 waves)
    constexpr int G = 4 * TN + NX;                       // MFMAs (gaps) per k-step
    constexpr int R = 4 + TN + (TAIL ? 1 : 0);           // fragment reads per k-step
    constexpr int NL = 8 + 2 * TN + (TAIL ? 1 : 0);      // staging loads per k-tile and wave (1 KB each)
    constexpr int SL0 = NL / 3 + (NL % 3 > 0 ? 1 : 0), SL1 = NL / 3 + (NL % 3 > 1 ? 1 : 0), SL2 = NL / 3;   // staging slots in k-steps 0, 1, 2
    static_assert(SL0 + SL1 + SL2 == NL, "slot split");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int h = lane >> 5, r32 = lane & 31;

    const int n_tiles = a.N / C::BN;
    const int total = q_tiles * n_tiles;
    const int gx = gridDim.x >> 3, xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;   // gridDim.x % 8 == 0
    auto tile_of = [&](int i) { return (i * 8 + xcd) * gx + wx; };
    int n_my = 0;
    while (tile_of(n_my) < total) ++n_my;
    if (n_my == 0) return;

    const int nk = a.K >> 6, nk1 = a.K1 >> 6;            // nk is even (host check)
    const long long sa = (long long)a.lda * 2, sw = (long long)a.K * 2;   // lda2 == lda (host check)

    // ---- per-lane address state
    const unsigned sm = lds_addr4(smem);
    unsigned rdA[4], rdW[4], rdX[4];                     // fragment read bases per k-step: lane part ^ (s << 5) (XOR swizzle of gemm256)
    {
        const unsigned f = (unsigned)(r32 >> 1) & 7u;
        const unsigned lp0 = (unsigned)r32 * 128u + ((((unsigned)h) ^ f) << 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned lp = lp0 ^ ((unsigned)s << 5);
            rdA[s] = sm + (unsigned)wr * 16384u + lp;
            rdW[s] = sm + (unsigned)C::WBase + (unsigned)wc * (unsigned)(TN * 4096) + lp;
            rdX[s] = sm + 32768u + lp;
        }
    }
    const unsigned lr = (unsigned)lane >> 3;
    const unsigned swz = (((unsigned)lane & 7u) ^ (lr >> 1)) << 4;
    const unsigned voa0 = lr * (unsigned)sa + swz, voa1 = voa0 ^ 64u;     // staging load offsets: rows 8 inst + lr of the wave's strip, even / odd inst
    const unsigned vow0 = lr * (unsigned)sw + swz, vow1 = vow0 ^ 64u;
    const unsigned lwa = sm + (unsigned)wave * 8192u + (unsigned)lane * 16u;                                     // staging write bases
    const unsigned lww = sm + (unsigned)C::WBase + (unsigned)wave * (unsigned)(C::BN * 32) + (unsigned)lane * 16u;
    const unsigned lwx = sm + 32768u + (unsigned)lane * 16u;

    // ---- load cursor: the k-tile the staging loads fetch (three ahead of the one being multiplied), uniform
    int lc_i = 0, lc_kt = 0, lc_tm = 0, lc_tn = 0;
    const char *lc_pa = nullptr, *lc_pw = nullptr, *lc_px = nullptr;
    unsigned vox = 0;
    auto lc_set_tile = [&]() {
        const int lin = tile_of(lc_i < n_my ? lc_i : n_my - 1);       // past the end: the last tile again (never used)
        lc_tm = lin / n_tiles;
        lc_tn = lin - lc_tm * n_tiles;
        lc_kt = 0;
        lc_pa = reinterpret_cast<const char*>(a.A) + ((long long)lc_tm * 256 + wave * 64) * sa;
        lc_pw = reinterpret_cast<const char*>(a.W) + ((long long)lc_tn * C::BN + wave * (C::BN / 4)) * sw;
        lc_px = reinterpret_cast<const char*>(a.A);
        if constexpr (TAIL) {
            long long row = (long long)tail_base + (long long)lc_tm * e_tail + ((int)lr < e_tail ? (int)lr : e_tail - 1);
            row = row < a.M ? row : a.M - 1;
            vox = (unsigned)row * (unsigned)sa + swz;
        }
    };
    auto lc_advance = [&]() {
        ++lc_kt;
        lc_pa += 128; lc_pw += 128; lc_px += 128;
        if (lc_kt == nk1 && nk1 < nk) {          // the concat-free second operand (skip_linear): k >= K1 comes from A2
            lc_pa = reinterpret_cast<const char*>(a.A2) + ((long long)lc_tm * 256 + wave * 64) * sa;
            lc_px = reinterpret_cast<const char*>(a.A2);
        }
        if (lc_kt == nk) { ++lc_i; lc_set_tile(); }
    };
    lc_set_tile();

    f32x4 sreg[2][NL];           // staging register sets: k-tile u lives in set u & 1
    auto stage_load = [&](auto idx_tag, f32x4& r) {
        constexpr int IDX = decltype(idx_tag)::value;
        if constexpr (IDX < 8) gload4(r, (IDX & 1) ? voa1 : voa0, lc_pa + (long long)(IDX * 8) * sa);
        else if constexpr (IDX < 8 + 2 * TN) gload4(r, ((IDX - 8) & 1) ? vow1 : vow0, lc_pw + (long long)((IDX - 8) * 8) * sw);
        else gload4(r, vox, lc_px);
    };
    // one staging slot: register IDX of set ST (k-tile u+1) has landed -> LDS stage ST; the register then receives k-tile u+3
    auto slot = [&](auto idx_tag, auto st_tag) {
        constexpr int IDX = decltype(idx_tag)::value, ST = decltype(st_tag)::value;
        f32x4& r = sreg[ST][IDX];
        wait_vm4<2 * NL - 1>();                 // younger than this register's load: the rest of its set, the other set, this k-tile's slots so far
        if constexpr (IDX < 8) lds_wr4<ST * kABytes + IDX * 1024>(lwa, r);
        else if constexpr (IDX < 8 + 2 * TN) lds_wr4<ST * C::WBytes + (IDX - 8) * 1024>(lww, r);
        else lds_wr4<ST * kABytes>(lwx, r);
        stage_load(idx_tag, r);
    };

    bf16x8 fa[2][4], fb[2][TN], fx[2];
    // fragment read number RI (of R) of k-step SN, stage ST, into buffer SN & 1
    auto frag_read = [&](auto ri_tag, auto sn_tag, auto st_tag) {
        constexpr int RI = decltype(ri_tag)::value, SN = decltype(sn_tag)::value, ST = decltype(st_tag)::value, B = SN & 1;
        // order: W0 A0 W1 A1 .. (the first MFMAs of a k-step need W0 A0 W1 ..), then the remaining A, then the tail rows
        if constexpr (RI < 2 * TN && RI < 8) {
            if constexpr ((RI & 1) == 0) lds_rd4<ST * C::WBytes + (RI / 2) * 4096>(fb[B][RI / 2], rdW[SN]);
            else lds_rd4<ST * kABytes + (RI / 2) * 4096>(fa[B][RI / 2], rdA[SN]);
        } else if constexpr (RI < 4 + TN) {
            constexpr int i = RI - TN;           // TN <= 4: reads 2 TN .. 3 + TN are A fragments TN .. 3
            lds_rd4<ST * kABytes + i * 4096>(fa[B][i], rdA[SN]);
        } else {
            lds_rd4<ST * kABytes>(fx[B], rdX[SN]);
        }
    };

    f32x16 acc[4][TN], accx[NX > 0 ? NX : 1];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) { acc[i][j] = zero16; asm volatile("" : "+a"(acc[i][j])); }
#pragma unroll
        for (int n = 0; n < NX; ++n) { accx[n] = zero16; asm volatile("" : "+a"(accx[n])); }
    };

    // ---- one k-tile: stage PAR is multiplied; set / stage 1 - PAR receives k-tile u+1 (and its registers k-tile u+3)
    auto body = [&](auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value, OTH = 1 - PAR;
        [&]<int... SI>(std::integer_sequence<int, SI...>) {
            ([&] {
                constexpr int s = SI, B = s & 1;
                constexpr int slots_before = s == 0 ? 0 : s == 1 ? SL0 : SL0 + SL1;
                constexpr int nslots = s == 0 ? SL0 : s == 1 ? SL1 : s == 2 ? SL2 : 0;
                constexpr int nops = R + nslots;
                [&]<int... GI>(std::integer_sequence<int, GI...>) {
                    ([&] {
                        constexpr int g = GI;
                        if constexpr (g == 0) {
                            // this k-step's fragments were read during the previous k-step, in front of its staging slots' LDS writes
                            if constexpr (s == 1) wait_lgkm4<SL0>();
                            else if constexpr (s == 2) wait_lgkm4<SL1>();
                            else wait_lgkm4<0>();
                            if constexpr (s == 3) __builtin_amdgcn_s_barrier();   // stage OTH complete, stage PAR free (see the header)
                        }
                        if constexpr (g < 4 * TN) {
                            mfma4(acc[g / TN][g % TN], fb[B][g % TN], fa[B][g / TN]);
                        } else {
                            // tail rows: the workgroup's 2 TN tail column tiles over 4 waves -- this wave's column span, tiles of parity wr
                            constexpr int n = g - 4 * TN;
                            constexpr int j0 = 2 * n, j1 = 2 * n + 1;                 // wr = 0 -> tile 2n, wr = 1 -> tile 2n + 1 (if it exists)
                            if (wr == 0) mfma4(accx[n], fb[B][j0], fx[B]);
                            else if constexpr (j1 < TN) mfma4(accx[n], fb[B][j1], fx[B]);
                        }
                        // the memory operations behind this MFMA: ops [lo, hi) of this k-step's list (reads first, then staging slots)
                        constexpr int per = (nops + G - 1) / G;
                        constexpr int lo = g * per < nops ? g * per : nops, hi = (g + 1) * per < nops ? (g + 1) * per : nops;
                        [&]<int... OI>(std::integer_sequence<int, OI...>) {
                            ([&] {
                                constexpr int op = lo + OI;
                                if constexpr (op < R) {
                                    if constexpr (s < 3) frag_read(std::integral_constant<int, op>{}, std::integral_constant<int, s + 1>{}, std::integral_constant<int, PAR>{});
                                    else frag_read(std::integral_constant<int, op>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, OTH>{});
                                } else {
                                    slot(std::integral_constant<int, slots_before + op - R>{}, std::integral_constant<int, OTH>{});
                                }
                            }(), ...);
                        }(std::make_integer_sequence<int, hi - lo>{});
                    }(), ...);
                }(std::make_integer_sequence<int, G>{});
            }(), ...);
        }(std::make_integer_sequence<int, 4>{});
        lc_advance();
    };

    // ---- epilogue of tile `lin`: lane = output row, register quad g of accumulator (i, j) = columns 32 j + 8 g + 4 h .. + 3
    constexpr bool HAS_BIAS = EPI != EPI_STORE;
    constexpr bool RESID = EPI == EPI_BIAS_RESID;
    constexpr bool WRITES_X = EPI == EPI_BIAS_RESID || EPI == EPI_BIAS_SET;
    auto epilogue = [&](int lin) {
        const int tm = lin / n_tiles, tn = lin - tm * n_tiles;
        const bool use_out = EPI != EPI_BIAS_SET && a.out != nullptr;
        const bool has_bias = HAS_BIAS && a.bias != nullptr;
        const int colw = tn * C::BN + wc * (32 * TN);                  // first column of this wave
        const long long row0 = (long long)tm * 256 + wr * 128 + r32;
        auto load_bias = [&](int j, f32x4 (&b)[4]) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = has_bias ? *reinterpret_cast<const f32x4*>(a.bias + colw + j * 32 + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
        };
        auto xptr = [&](int i, int j, int g) {
            return reinterpret_cast<f32x4*>(a.xres + (row0 + i * 32) * a.N + colw + j * 32 + 8 * g + 4 * h);
        };
        // units u = j * 4 + i (column tile outer: its bias quads serve four row slabs).  No global load may wait behind a store
        // (vmcnt retires in order): the residual quads of unit u + 1 and the bias quads of column tile j + 1 are requested BEFORE
        // the stores of unit u.
        f32x4 bias[2][4], xl[2][4];
        load_bias(0, bias[0]);
        if constexpr (RESID) {
#pragma unroll
            for (int g = 0; g < 4; ++g) xl[0][g] = *xptr(0, 0, g);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int u = j * 4 + i;
                if constexpr (RESID) {
                    if (u + 1 < 4 * TN) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) xl[(u + 1) & 1][g] = *xptr((u + 1) & 3, (u + 1) >> 2, g);
                    }
                }
                if (i == 3 && j + 1 < TN) load_bias(j + 1, bias[(j + 1) & 1]);
                const f32x16 t = acc[i][j];
                uint2 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 q = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
                    if constexpr (HAS_BIAS) q += bias[j & 1][g];
                    if constexpr (EPI == EPI_BIAS_GELU) q = gelu4(q);
                    if constexpr (RESID) q = xl[u & 1][g] + q;
                    if constexpr (WRITES_X) *xptr(i, j, g) = q;
                    v[g] = pack4_bf16(q);
                }
                if (use_out) {
                    // 16-byte row segments built in registers (v_permlane32_swap pairs the lane halves, as gemm256's epilogue);
                    // head-major qkv: a 32-column block lies inside one (q | k | v, head) unit
                    const long long orow = row0 + i * 32;
                    const int ocol = colw + j * 32;
                    bf16_t* obase = a.hm.L ? a.out + hm_offset(a.hm, (int)orow, ocol) : a.out + orow * a.ldo + ocol;
#pragma unroll
                    for (int gp = 0; gp < 4; gp += 2) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(v[gp].x, v[gp + 1].x, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(v[gp].y, v[gp + 1].y, false, false);
                        const uint4 o = {s0[0], s1[0], s0[1], s1[1]};
                        *reinterpret_cast<uint4*>(obase + 8 * gp + 8 * h) = o;
                    }
                }
            }
        }
        if constexpr (TAIL) {
            // tail rows: lane = tail row index (valid below e_tail), accx[n] = column tile 2 n + wr of this wave's span
            const long long row = (long long)tail_base + (long long)tm * e_tail + r32;
            const bool row_ok = r32 < e_tail && row < a.M;
#pragma unroll
            for (int n = 0; n < NX; ++n) {
                const int j = 2 * n + wr;
                if (j >= TN) continue;
                const f32x16 t = accx[n];
                if (row_ok) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int col = colw + j * 32 + 8 * g + 4 * h;
                        f32x4 q = {t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
                        if (has_bias) q += *reinterpret_cast<const f32x4*>(a.bias + col);
                        if constexpr (EPI == EPI_BIAS_GELU) q = gelu4(q);
                        if constexpr (WRITES_X) {
                            f32x4* xp = reinterpret_cast<f32x4*>(a.xres + row * a.N + col);
                            if constexpr (RESID) q = *xp + q;
                            *xp = q;
                        }
                        if (use_out) {
                            const long long off = a.hm.L ? hm_offset(a.hm, (int)row, col) : row * a.ldo + col;
                            *reinterpret_cast<uint2*>(a.out + off) = pack4_bf16(q);
                        }
                    }
                }
            }
        }
    };

    // ---- prologue: k-tiles 0 and 1 requested, k-tile 0 written to stage 0 (its registers then fetch k-tile 2)
    [&]<int... I>(std::integer_sequence<int, I...>) { (stage_load(std::integral_constant<int, I>{}, sreg[0][I]), ...); }(std::make_integer_sequence<int, NL>{});
    lc_advance();
    [&]<int... I>(std::integer_sequence<int, I...>) { (stage_load(std::integral_constant<int, I>{}, sreg[1][I]), ...); }(std::make_integer_sequence<int, NL>{});
    lc_advance();
    [&]<int... I>(std::integer_sequence<int, I...>) { (slot(std::integral_constant<int, I>{}, std::integral_constant<int, 0>{}), ...); }(std::make_integer_sequence<int, NL>{});
    lc_advance();
    wait_lgkm4<0>();
    __builtin_amdgcn_s_barrier();
    [&]<int... RI>(std::integer_sequence<int, RI...>) {
        (frag_read(std::integral_constant<int, RI>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}), ...);
    }(std::make_integer_sequence<int, R>{});
    zero_acc();

    int ci = 0, ckt = 0;
    const int nkt_total = n_my * nk;
    for (int u = 0; u < nkt_total; u += 2) {
        body(std::integral_constant<int, 0>{});
        body(std::integral_constant<int, 1>{});
        ckt += 2;
        if (ckt == nk) {
            // hipcc does not know the asm statements are MFMAs: drain the pipe before its own reads of the accumulators, and make
            // sure the prefetched fragments of the next k-tile have landed before compiler-generated code could move them
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
#pragma unroll
            for (int n = 0; n < NX; ++n) asm volatile("" : "+a"(accx[n]));
            epilogue(tile_of(ci));
            ++ci;
            ckt = 0;
            zero_acc();
        }
    }
    wait_vm4<0>();      // the run-ahead loads of the cursor past the last tile
}

template <int TN>
bool plan4(int M, int N, int num_cus, int& q_out, int& e_out, double& cost_out) {
    if (N % (64 * TN) || M < 256) return false;
    const int nt = N / (64 * TN);
    const int q_hi = M / 256, q_lo = (M + 263) / 264;
    double best = 1e30;
    int best_q = -1, best_e = 0;
    for (int q = q_hi; q >= q_lo && q >= 1; --q) {
        const int tail = M - 256 * q;
        const int e = tail > 0 ? (tail + q - 1) / q : 0;
        if (e > 8) continue;
        const long long tiles = (long long)q * nt;
        const double rounds = (double)((tiles + num_cus - 1) / num_cus);
        const double per_tile = (double)TN * (e > 0 ? (4.0 * TN + (TN + 1) / 2) / (4.0 * TN) : 1.0);   // MFMA work of a tile, in units of 256 x 64
        const double cost = rounds * per_tile;
        if (cost < best - 1e-9) { best = cost; best_q = q; best_e = e; }
    }
    if (best_q < 0) return false;
    q_out = best_q; e_out = best_e; cost_out = best;
    return true;
}

template <int TN, bool TAIL>
hipError_t launch4(const GemmArgs<bf16_t>& a, int epi, int q, int e, int num_cus, hipStream_t s) {
    const int tiles = q * (a.N / (64 * TN));
    int grid = num_cus;
    if (tiles < grid) grid = (tiles + 7) / 8 * 8;
    const int tail_base = 256 * q;
#define DD_L4(E)                                                                                                             \
    {                                                                                                                        \
        hipLaunchKernelGGL((gemm4w_kernel<TN, E, TAIL>), dim3(grid), dim3(256), G4<TN>::Lds, s, a, q, e, tail_base);          \
        return hipGetLastError();                                                                                            \
    }
    switch (epi) {
        case EPI_STORE: DD_L4(EPI_STORE)
        case EPI_BIAS_GELU: DD_L4(EPI_BIAS_GELU)
        case EPI_BIAS_RESID: DD_L4(EPI_BIAS_RESID)
        case EPI_BIAS_SET: DD_L4(EPI_BIAS_SET)
        case EPI_BIAS_STORE: DD_L4(EPI_BIAS_STORE)
    }
#undef DD_L4
    return hipErrorInvalidValue;
}

template <int TN, bool TAIL>
hipError_t init4() {
    hipError_t e = hipSuccess;
#define DD_A4(E)                                                                                                     \
    if (e == hipSuccess)                                                                                             \
        e = hipFuncSetAttribute((const void*)gemm4w_kernel<TN, E, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, G4<TN>::Lds);
    DD_A4(EPI_STORE) DD_A4(EPI_BIAS_GELU) DD_A4(EPI_BIAS_RESID) DD_A4(EPI_BIAS_SET) DD_A4(EPI_BIAS_STORE)
#undef DD_A4
    return e;
}

}  // namespace

hipError_t init_gemm4w_kernels() {
    hipError_t e = init4<3, true>();
    if (e == hipSuccess) e = init4<3, false>();
    if (e == hipSuccess) e = init4<2, true>();
    if (e == hipSuccess) e = init4<2, false>();
    return e;
}

// Shapes the kernel takes: bf16, K a multiple of 128 (an even number of k-tiles) of at least 256, K1 a multiple of 64, both A operands
// with one row stride, N a multiple of 192 or 128, M >= 256, 32-bit lane offsets.  The embed_dim 512 models keep gemm256 on their
// fallback paths (their product path has no plain GEMM).
bool gemm4w_supported(const GemmArgs<bf16_t>& a) {
    if (a.K % 128 || a.K < 256 || a.K1 % 64 || a.K1 > a.K || (a.K1 < a.K && (!a.A2 || a.lda != a.lda2))) return false;
    if (a.M < 256 || (a.N % 192 && a.N % 128) || a.N < 768) return false;
    if ((long long)a.M * a.lda * 2 >= (1ll << 32) || (long long)a.K * 2 * 8 >= (1ll << 31)) return false;
    return true;
}

hipError_t launch_gemm4w(const GemmArgs<bf16_t>& a, int epilogue, int num_cus, hipStream_t s) {
    if (!gemm4w_supported(a)) return hipErrorInvalidValue;
    const int cus = num_cus >= 8 ? num_cus / 8 * 8 : 256;
    int q3 = 0, e3 = 0, q2 = 0, e2 = 0;
    double c3 = 1e30, c2 = 1e30;
    const bool ok3 = plan4<3>(a.M, a.N, cus, q3, e3, c3), ok2 = plan4<2>(a.M, a.N, cus, q2, e2, c2);
    if (!ok3 && !ok2) return hipErrorInvalidValue;
    // (a 256 x 128 tile moves 1.3 x the operand bytes of a 256 x 192 one per flop: it must win on rounds to be chosen)
    if (ok3 && (!ok2 || c3 <= c2 * 1.15)) return e3 > 0 ? launch4<3, true>(a, epilogue, q3, e3, cus, s) : launch4<3, false>(a, epilogue, q3, e3, cus, s);
    return e2 > 0 ? launch4<2, true>(a, epilogue, q2, e2, cus, s) : launch4<2, false>(a, epilogue, q2, e2, cus, s);
}

}  // namespace dd
