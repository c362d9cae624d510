// Probe (development tool, not product): how much a CU takes in through vector-memory instructions (LDS-DMA global_load_lds_dwordx4 and plain
// global_load_dwordx4, 1 KB per wave-instruction), and what such an instruction costs the MFMA stream of ANOTHER wave on the same SIMD -- the pattern of every weight-streaming kernel here (fused block tail, gemm256, rowlin, qkv_attention).
//   hipcc -O3 -std=c++20 --offload-arch=gfx950 tools/experiments/dma_mfma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
// One 256-thread workgroup per CU (128 KB of LDS), every workgroup streams the SAME 4 MB buffer (L2-shared, like a weight image) `reps` times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <utility>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// plain global_load_dwordx4 into a register ring of DEPTH quads, each consumed by a ds_write_b128 when it lands; NWAVES waves per workgroup all load
template <int DEPTH, int NWAVES, bool LDSDMA>
__global__ void __launch_bounds__(NWAVES * 64) gl_probe(const char* __restrict__ w, float* out, int steps, long long wbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = w + lane * 16;
    const long long span = wbytes - 2048;
    typedef __attribute__((ext_vector_type(4))) float f4;
    f4 r[DEPTH];
    const unsigned lds = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + wave * 8192 + lane * 16;
    auto addr = [&](int i) { return base + ((((unsigned)i * NWAVES + wave) * 1024u) & ((2u << 20) - 1)); };
    if constexpr (LDSDMA) {
        for (int i = 0; i < DEPTH; ++i) __builtin_amdgcn_global_load_lds((gptr_t)addr(i), (lptr_t)(smem + wave * 8192 + (i & 7) * 1024), 16, 0, 0);
        for (int i = 0; i < steps; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DEPTH - 1) : "memory");
            __builtin_amdgcn_global_load_lds((gptr_t)addr(i + DEPTH), (lptr_t)(smem + wave * 8192 + (i & 7) * 1024), 16, 0, 0);
        }
    } else {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[d]) : "v"(addr(d)) : "memory");
        for (int i = 0; i < steps; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                asm volatile("s_waitcnt vmcnt(%2)\n\tds_write_b128 %1, %0" : "+v"(r[d]) : "v"(lds + (d & 7) * 1024), "i"(DEPTH - 1) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[d]) : "v"(addr(i + d + DEPTH)) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (steps < 0) out[threadIdx.x] = r[0][0];
}
// roles: 8 waves per CU (two per SIMD); waves 0-3 issue MF MFMAs per step and never touch memory, waves 4-7 issue one LDS-DMA piece per step and
// never an MFMA (ROLE 0); ROLE 1: only the MFMA waves work; ROLE 2: only the DMA waves work.  Independent pipes: time(0) = max(time(1), time(2)).
template <int MF, int ROLE, int DMODE = 0>
__global__ void __launch_bounds__(512) role_probe(const char* __restrict__ w, float* out, int steps, long long wbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = w + lane * 16;
    const unsigned mask = (2u << 20) - 1;          // the stream wraps in 2 MB (32-bit offsets, no division in the loop)
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
    if (wave < 4) {
        if (ROLE != 2)
            for (int st = 0; st < steps; ++st)
#pragma unroll
                for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
    } else if (ROLE != 1) {
        // ADD_TID_ENABLE (word 3 bit 23) + stride 16: the hardware adds lane * 16 itself, the instruction carries no VGPR address
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 16, (int)(wbytes / 16), 0x00020000 | (1 << 23));
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)wbytes, 0x00020000);   // raw buffer: SGPR base + scalar offset + 32-bit lane offset
        typedef __attribute__((ext_vector_type(4))) float f4;
        f4 r[8];
        auto one = [&](int i, int slot) {
            const unsigned off = (((unsigned)i * 4 + wave) * 1024u) & mask;
            if constexpr (DMODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(smem + wave * 8192 + slot * 1024), 16, 0, 0);
            else if constexpr (DMODE == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + wave * 8192 + slot * 1024), 16, 0, (int)off, 0, 0);
            else if constexpr (DMODE == 3) __builtin_amdgcn_global_load_lds((gptr_t)(w + lane * 4 + off), (lptr_t)(smem + wave * 8192 + slot * 256), 4, 0, 0);
            else if constexpr (DMODE == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, (lptr_t)(smem + wave * 8192 + slot * 1024), 16, lane * 16, (int)off, 0, 0);
            else if constexpr (DMODE == 5) {
                const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + wave * 8192 + slot * 1024);
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(lane * 16), "s"(w + off) : "memory", "m0");
            }
        };
        if constexpr (DMODE == 1) {
#pragma unroll
            for (int d = 0; d < 8; ++d) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[d]) : "v"(base + ((((unsigned)d * 4 + wave) * 1024u) & mask)) : "memory");
            for (int i = 0; i < steps; i += 8) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    asm volatile("s_waitcnt vmcnt(7)" : "+v"(r[d])::"memory");
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[d]) : "v"(base + ((((unsigned)(i + d + 8) * 4 + wave) * 1024u) & mask)) : "memory");
                }
            }
        } else {
            for (int i = 0; i < 8; ++i) one(i, i & 7);
            for (int i = 0; i < steps; ++i) {
                asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                one(i + 8, i & 7);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}
// the same wave issues one piece and then MF MFMAs per step (4 waves per CU, one per SIMD): the 1-wave-per-SIMD kernels' pattern
template <int MF, int DMODE>
__global__ void __launch_bounds__(256) self_probe(const char* __restrict__ w, float* out, int steps, long long wbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = w + lane * 16;
    const unsigned mask = (2u << 20) - 1;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    const bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {1, 1, 1, 1, 1, 1, 1, 1};
    auto one = [&](int i, int slot) {
        const unsigned off = (((unsigned)i * 4 + wave) * 1024u) & mask;
        if constexpr (DMODE == 0) __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(smem + wave * 8192 + slot * 1024), 16, 0, 0);
        else {
            const unsigned la = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + wave * 8192 + slot * 1024);
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(lane * 16), "s"(w + off) : "memory", "m0");
        }
    };
    for (int i = 0; i < 8; ++i) one(i, i & 7);
    for (int i = 0; i < steps; ++i) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        one(i + 8, i & 7);
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[m & 3], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int MF, int DMODE>
int run_self(const char* w, float* out, long long wbytes, int cus) {
    const int steps = 8192;
    auto k = self_probe<MF, DMODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 128 * 1024, 0, w, out, 256, wbytes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 128 * 1024, 0, w, out, steps, wbytes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("same wave: one 1 KB LDS-DMA piece (%s) + %d MFMA per step, one wave per SIMD: %8.3f ms = %5.0f cycles per step at 2.1 GHz (the MFMAs alone: %d)\n",
           DMODE == 0 ? "64-bit lane addresses" : "SGPR base + 32-bit lane offset", MF, ms, ms * 1e-3 * 2.1e9 / steps, MF * 32);
    fflush(stdout);
    return 0;
}

// rowlin / fused-tail gap pattern in one wave per SIMD: per step DMA pieces (0, 1 or 2) + MF x [wait: fragment 8 gaps old; MFMA; ds_read_b128 of the
// fragment 8 gaps ahead (READS) ] as one asm statement per gap
template <int MF, int NDMA, bool READS>
__global__ void __launch_bounds__(256) gap_probe(const char* __restrict__ w, float* out, int steps, long long wbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = w + lane * 16;
    const unsigned mask = (2u << 20) - 1;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 fr[8];
    const bf16x8 fb = {1, 1, 1, 1, 1, 1, 1, 1};
    const unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
    for (int i = 0; i < 8; ++i) fr[i] = fb;
    auto one = [&](int i, int slot) {
        const unsigned off = (((unsigned)i * 4 + wave) * 1024u) & mask;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(smem + 65536 + wave * 8192 + slot * 1024), 16, 0, 0);
    };
    for (int i = 0; i < 8; ++i) one(i, i & 7);
    for (int i = 0; i < steps; ++i) {
        if (NDMA > 0) { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); one(2 * i + 8, i & 7); }
        if (NDMA > 1) { asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); one(2 * i + 9, (i + 1) & 7); }
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            if (READS)
                asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%4"
                             : "+v"(acc[m & 3]), "+v"(fr[m & 7]) : "v"(fb), "v"(la), "i"((m & 31) * 1024));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(fr[m & 7]), "v"(fb));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int MF, int NDMA, bool READS>
int run_gap(const char* w, float* out, long long wbytes, int cus) {
    const int steps = 8192;
    auto k = gap_probe<MF, NDMA, READS>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 128 * 1024, 0, w, out, 256, wbytes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 128 * 1024, 0, w, out, steps, wbytes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("gaps: per step %d LDS-DMA piece(s) + %d x [MFMA%s], one wave per SIMD: %8.3f ms = %5.0f cycles per step at 2.1 GHz (MFMA pipe alone: %d)\n", NDMA, MF,
           READS ? " + ds_read_b128 of a fresh fragment" : "", ms, ms * 1e-3 * 2.1e9 / steps, MF * 32);
    fflush(stdout);
    return 0;
}

// rowlin's k-step: 24 gaps [MFMA + ds_read_b128], NDMA pieces per wave spread over the gaps, requested AHEAD k-steps ahead; at gap 2 every wave
// waits for its pieces of the NEXT k-step (vmcnt) and joins a workgroup barrier (BAR) -- the hand-off of an LDS ring slot
template <int NDMA, int AHEAD, bool BAR>
__global__ void __launch_bounds__(256) kstep_probe(const char* __restrict__ w, float* out, int steps, long long wbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = w + lane * 16;
    const unsigned mask = (4u << 20) - 1;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 fr[8];
    const bf16x8 fb = {1, 1, 1, 1, 1, 1, 1, 1};
    const unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
    for (int i = 0; i < 8; ++i) fr[i] = fb;
    auto one = [&](int ks, int j) {      // piece j of this wave for k-step ks: the stream is [k-step][wave][piece]
        const unsigned off = ((((unsigned)ks * 4 + wave) * NDMA + j) * 1024u) & mask;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + off), (lptr_t)(smem + 32768 + (ks & 3) * 24576 + (wave * NDMA + j) % 24 * 1024), 16, 0, 0);
    };
    for (int ks = 0; ks < AHEAD; ++ks) for (int j = 0; j < NDMA; ++j) one(ks, j);
    for (int ks = 0; ks < steps; ++ks) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (m == 2) {
                if (BAR) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"((AHEAD - 1) * NDMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1) * NDMA) : "memory");
            }
            if (m >= 3 && (m - 3) % 3 == 0 && (m - 3) / 3 < NDMA) one(ks + AHEAD, (m - 3) / 3);
            asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%4"
                         : "+v"(acc[m & 3]), "+v"(fr[m & 7]) : "v"(fb), "v"(la), "i"(m * 1024));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int NDMA, int AHEAD, bool BAR>
int run_kstep(const char* w, float* out, long long wbytes, int cus) {
    const int steps = 2048;
    auto k = kstep_probe<NDMA, AHEAD, BAR>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 144 * 1024, 0, w, out, 64, wbytes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 144 * 1024, 0, w, out, steps, wbytes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k-step: 24 x [MFMA + ds_read_b128] + %d LDS-DMA pieces per wave, %d k-steps ahead, %s: %8.3f ms = %5.0f cycles per k-step at 2.1 GHz (MFMA pipe alone: 768)\n", NDMA, AHEAD,
           BAR ? "vmcnt + workgroup barrier per k-step" : "vmcnt wait only", ms, ms * 1e-3 * 2.1e9 / steps);
    fflush(stdout);
    return 0;
}

template <bool AG> __device__ __forceinline__ void pin_acc(f32x16& y) { if constexpr (AG) asm volatile("" : "+a"(y)); else asm volatile("" : "+v"(y)); }
template <bool AG, int OFF> __device__ __forceinline__ void gap_acc(f32x16& acc, bf16x8& fr, const bf16x8& fb, unsigned la) {
    if constexpr (AG) asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%4" : "+a"(acc), "+v"(fr) : "v"(fb), "v"(la), "i"(OFF));
    else asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tds_read_b128 %1, %3 offset:%4" : "+v"(acc), "+v"(fr) : "v"(fb), "v"(la), "i"(OFF));
}
// rowlin's accumulator set: 24 independent 32 x 32 tiles per wave (384 registers), the first NA of them in AGPRs; no requests, barrier per k-step
template <int NA>
__global__ void __launch_bounds__(256) acc_probe(const char* __restrict__ w, float* out, int steps, long long wbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[24];
    for (int i = 0; i < 24; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 fr[8];
    const bf16x8 fb = {1, 1, 1, 1, 1, 1, 1, 1};
    const unsigned la = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
    for (int i = 0; i < 8; ++i) fr[i] = fb;
    [&]<int... T>(std::integer_sequence<int, T...>) { (pin_acc<(T < NA)>(acc[T]), ...); }(std::make_integer_sequence<int, 24>{});
    for (int ks = 0; ks < steps; ++ks) {
        [&]<int... M>(std::integer_sequence<int, M...>) {
            (((M == 2 ? (void)__builtin_amdgcn_s_barrier() : (void)0), gap_acc<(M < NA), M * 1024>(acc[M], fr[M & 7], fb, la)), ...);
        }(std::make_integer_sequence<int, 24>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    [&]<int... T>(std::integer_sequence<int, T...>) { (pin_acc<(T < NA)>(acc[T]), ...); }(std::make_integer_sequence<int, 24>{});
    float s = 0.f;
    for (int i = 0; i < 24; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int NA>
int run_acc(const char* w, float* out, long long wbytes, int cus) {
    const int steps = 2048;
    auto k = acc_probe<NA>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 144 * 1024, 0, w, out, 64, wbytes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(cus), dim3(256), 144 * 1024, 0, w, out, steps, wbytes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k-step, 24 independent accumulator tiles (%d in AGPRs), 24 x [MFMA + ds_read_b128], barrier per k-step: %8.3f ms = %5.0f cycles per k-step at 2.1 GHz (MFMA pipe alone: 768)\n", NA, ms,
           ms * 1e-3 * 2.1e9 / steps);
    fflush(stdout);
    return 0;
}

template <int MF, int ROLE, int DMODE = 0>
int run_role(const char* w, float* out, long long wbytes, int cus) {
    const int steps = 8192;
    auto k = role_probe<MF, ROLE, DMODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(cus), dim3(512), 128 * 1024, 0, w, out, 256, wbytes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(cus), dim3(512), 128 * 1024, 0, w, out, steps, wbytes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("roles (dma mode %d: 0 global_load_lds x4, 1 global_load_dwordx4 -> VGPR, 3 global_load_lds dword, 4 buffer_load_dwordx4 lds (32-bit lane offset), 5 global_load_lds x4 saddr + 32-bit lane offset): %d MFMA per step on waves 0-3, one piece per step on waves 4-7, %s: %8.3f ms = %5.0f cycles per step at 2.1 GHz\n", DMODE, MF,
           ROLE == 0 ? "both" : ROLE == 1 ? "MFMA waves only" : "DMA waves only", ms, ms * 1e-3 * 2.1e9 / steps);
    fflush(stdout);
    return 0;
}

template <int DEPTH, int NWAVES, bool LDSDMA>
int run_gl(const char* w, float* out, long long wbytes, int cus) {
    const int steps = 8192 * 4 / NWAVES;             // 1 KB per step and wave: 32 MB per CU
    auto k = gl_probe<DEPTH, NWAVES, LDSDMA>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(cus), dim3(NWAVES * 64), 128 * 1024, 0, w, out, 256, wbytes);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(cus), dim3(NWAVES * 64), 128 * 1024, 0, w, out, steps, wbytes);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = (double)steps * NWAVES * 1024 / (ms * 1e-3) * 1e-9;
    printf("%s, %d waves per CU, %2d in flight per wave: %8.3f ms  %6.1f GB/s per CU  (%5.1f per wave)\n", LDSDMA ? "global_load_lds_dwordx4" : "global_load_dwordx4 + ds_write_b128",
           NWAVES, DEPTH, ms, gbs, gbs / NWAVES);
    fflush(stdout);
    return 0;
}

int main() {
    const long long wbytes = 64ll << 20;
    char* w; float* out;
    CK(hipMalloc((void**)&w, wbytes)); CK(hipMemset(w, 1, wbytes)); CK(hipMalloc((void**)&out, 4096));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("CUs %d\n", cus);
    const long long shared = 4ll << 20;   // every workgroup streams the same 2 MB (an XCD's L2 holds it)
    if (run_role<5, 1>(w, out, shared, cus) || run_role<5, 2>(w, out, shared, cus) || run_role<5, 0>(w, out, shared, cus)) return 1;
    if (run_role<10, 1>(w, out, shared, cus) || run_role<10, 0>(w, out, shared, cus)) return 1;
    if (run_role<5, 2, 1>(w, out, shared, cus) || run_role<5, 0, 1>(w, out, shared, cus)) return 1;
    if (run_role<5, 2, 3>(w, out, shared, cus) || run_role<5, 0, 3>(w, out, shared, cus)) return 1;
    if (run_role<5, 2, 4>(w, out, shared, cus) || run_role<5, 0, 4>(w, out, shared, cus)) return 1;
    if (run_role<5, 2, 5>(w, out, shared, cus) || run_role<5, 0, 5>(w, out, shared, cus)) return 1;
    if (run_self<0, 0>(w, out, shared, cus) || run_self<0, 5>(w, out, shared, cus) || run_self<2, 0>(w, out, shared, cus) || run_self<2, 5>(w, out, shared, cus)) return 1;
    if (run_self<4, 0>(w, out, shared, cus) || run_self<4, 5>(w, out, shared, cus) || run_self<8, 0>(w, out, shared, cus) || run_self<8, 5>(w, out, shared, cus)) return 1;
    if (run_gap<4, 0, false>(w, out, shared, cus) || run_gap<4, 0, true>(w, out, shared, cus) || run_gap<4, 1, false>(w, out, shared, cus) || run_gap<4, 1, true>(w, out, shared, cus)) return 1;
    if (run_gap<4, 2, true>(w, out, shared, cus) || run_gap<8, 0, true>(w, out, shared, cus) || run_gap<8, 2, true>(w, out, shared, cus)) return 1;
    if (run_acc<0>(w, out, shared, cus) || run_acc<16>(w, out, shared, cus) || run_acc<24>(w, out, shared, cus)) return 1;
    if (run_kstep<0, 1, false>(w, out, shared, cus) || run_kstep<0, 1, true>(w, out, shared, cus)) return 1;
    if (run_kstep<7, 3, false>(w, out, shared, cus) || run_kstep<7, 3, true>(w, out, shared, cus) || run_kstep<7, 2, true>(w, out, shared, cus) || run_kstep<7, 1, true>(w, out, shared, cus)) return 1;
    if (run_kstep<4, 3, true>(w, out, shared, cus)) return 1;
#define RG(DE, NW, LD) if (run_gl<DE, NW, LD>(w, out, shared, cus)) return 1;
    RG(1, 1, true) RG(2, 1, true) RG(4, 1, true) RG(8, 1, true) RG(16, 1, true) RG(32, 1, true)
    RG(8, 4, true) RG(8, 8, true) RG(16, 8, true) RG(8, 16, true)
    RG(1, 1, false) RG(2, 1, false) RG(4, 1, false) RG(8, 1, false) RG(16, 1, false)
    RG(8, 4, false) RG(8, 8, false) RG(16, 8, false) RG(8, 16, false)
    return 0;
}
