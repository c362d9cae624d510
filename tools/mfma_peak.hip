// Dev tool: sustained MFMA rate of v_mfma_f32_32x32x16_bf16 on this device (no memory traffic at all).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 12345.f) out[0] = s;
}

// random operands (sign + mantissa random, exponent of 1.0): realistic toggle rate in the multiplier arrays
__global__ void __launch_bounds__(256) mfma_loop_random(float* out, int iters) {
    f32x16 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[4], rb[4];
    unsigned h = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + 12345u;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
            ra[i][e] = (h & 0x807f807fu) | 0x3f803f80u;
            h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
            rb[i][e] = (h & 0x807f807fu) | 0x3f803f80u;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, rb[j]), acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0];
    if (s == 12345.f) out[0] = s;
}

void run_random(int iters) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop_random, dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop_random, dim3(256), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("random operands 16 acc iters=%d: %.3f ms  %.1f TFLOP/s\n", iters, ms, 256.0 * 4 * iters * 16 * 32768.0 / ms / 1e9);
    }
    hipFree(d);
}

// short kernels launched back to back (the GEMMs of the sampler run ~50-100 us each)
void run_random_burst(int iters, int launches) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop_random, dim3(256), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(mfma_loop_random, dim3(256), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("random operands burst: %d launches x %d iters: %.1f us/launch  %.1f TFLOP/s\n", launches, iters, ms * 1e3 / launches,
               256.0 * 4 * iters * 16 * 32768.0 * launches / ms / 1e9);
    }
    hipFree(d);
}

// co-execution probe: NV independent packed-fp32 FMAs issued after every MFMA (MODE 0: MFMA only, 1: both, 2: VALU only)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE, int NV>
__global__ void __launch_bounds__(256) coexec(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)((threadIdx.x * 7 + e) & 15); b[e] = (__bf16)(float)((threadIdx.x + e * 3) & 7); }
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + i, 1.0f};
    const f32x2 m = {0.999f, 1.001f}, c = {0.5f, -0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE != 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[(i + k) & 7] = __builtin_elementwise_fma(v[(i + k) & 7], m, c);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i][0] + v[i][1];
    if (s == 12345.f) out[0] = s;
}
// same probe with plain (unpacked) v_fma_f32
template <int MODE, int NV>
__global__ void __launch_bounds__(256) coexec1(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)((threadIdx.x * 7 + e) & 15); b[e] = (__bf16)(float)((threadIdx.x + e * 3) & 7); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE != 2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            if (MODE != 0) {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[(i + k) & 7] = __builtin_fmaf(v[(i + k) & 7], 0.999f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i];
    if (s == 12345.f) out[0] = s;
}
template <int MODE, int NV>
void run_coexec1(int iters, const char* name) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((coexec1<MODE, NV>), dim3(512), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((coexec1<MODE, NV>), dim3(512), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("coexec1 %-27s NV=%d: %.3f ms\n", name, NV, ms);
    hipFree(d);
}

template <int MODE, int NV>
void run_coexec(int iters, const char* name) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((coexec<MODE, NV>), dim3(512), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((coexec<MODE, NV>), dim3(512), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("coexec %-28s NV=%d: %.3f ms\n", name, NV, ms);
    hipFree(d);
}

template <int NACC>
void run(int waves_per_simd, int iters, const char* name) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * 4 * iters * NACC * 32768.0;
        printf("%s waves/SIMD=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", name, waves_per_simd, iters, ms, flops / ms / 1e9);
    }
    hipFree(d);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    run<1>(1, iters * 4, "1 acc (dependent chain)");
    run<2>(1, iters * 2, "2 acc");
    run<1>(2, iters * 4, "1 acc (dependent chain)");
    run<2>(2, iters * 2, "2 acc");
    run<4>(1, iters, "4 acc");
    run<8>(1, iters, "8 acc");
    run<8>(2, iters, "8 acc");
    run<16>(1, iters / 2, "16 acc");
    run<8>(2, iters * 10, "8 acc long");
    run_random(iters / 2);
    run_random(iters * 5);
    run_coexec<0, 4>(iters, "MFMA only (2 waves/SIMD)");
    run_coexec<1, 4>(iters, "MFMA + 4 pk_fma each");
    run_coexec<2, 4>(iters, "4 pk_fma only");
    run_coexec<1, 6>(iters, "MFMA + 6 pk_fma each");
    run_coexec<2, 6>(iters, "6 pk_fma only");
    run_coexec1<1, 4>(iters, "MFMA + 4 v_fma each");
    run_coexec1<2, 4>(iters, "4 v_fma only");
    run_coexec1<1, 7>(iters, "MFMA + 7 v_fma each");
    run_coexec1<2, 7>(iters, "7 v_fma only");
    run_random_burst(128, 20);    // ~35 us of MFMA per launch at 1.9 PF
    run_random_burst(128, 200);
    run_random_burst(32, 200);    // = one 256x256x512 tile stream of 4 tiles per CU
    return 0;
}
