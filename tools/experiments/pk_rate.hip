// Probe (development tool, not product): VALU issue cost of v_pk_fma_f32 / v_pk_mul_f32 against v_fma_f32 for a lone wave per SIMD,
// dependent chains and independent ones, in cycles per instruction (s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/pk_rate.hip -o tools/bin/pk_rate && tools/bin/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long* out, float* sink, int steps) {
    f32x2 a = {1.f, 2.f}, b = {0.5f, 0.25f}, c = {3.f, 4.f}, d = {5.f, 6.f}, e = {1.f, 1.f};
    float x = 1.f, y = 2.f, z = 3.f, w = 4.f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z)); }
            if constexpr (MODE == 1) { asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x), "+v"(w) : "v"(y), "v"(z)); }
            if constexpr (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c)); }
            if constexpr (MODE == 3) { asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3" : "+v"(a), "+v"(d) : "v"(b), "v"(c)); }
            if constexpr (MODE == 4) { asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n\tv_pk_fma_f32 %1, %1, %3, %4\n\tv_pk_fma_f32 %2, %2, %3, %4" : "+v"(a), "+v"(d), "+v"(e) : "v"(b), "v"(c)); }
            if constexpr (MODE == 5) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
            if constexpr (MODE == 6) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(c)); }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[MODE] = t1 - t0;
    float sum = a[0] + a[1] + d[0] + d[1] + e[0] + e[1] + x + w;
    if (sum == 12.345f) sink[0] = sum;
}
int main() {
    unsigned long long* out; float* sink;
    if (hipMalloc((void**)&out, 64) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess) return 1;
    const int steps = 2000;
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, sink, steps);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, sink, steps);
    hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, sink, steps);
    hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, sink, steps);
    hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, out, sink, steps);
    hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, out, sink, steps);
    hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, out, sink, steps);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    unsigned long long h[8];
    if (hipMemcpy(h, out, 56, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* names[] = {"v_fma_f32 dependent", "v_fma_f32 x2 independent", "v_pk_fma_f32 dependent", "v_pk_fma_f32 x2 independent", "v_pk_fma_f32 x3 independent", "v_pk_mul_f32 dependent", "v_pk_fma_f32 dependent, sgpr src"};
    const int per[] = {1, 2, 1, 2, 3, 1, 1};
    for (int m = 0; m < 7; ++m) printf("%-34s %6.2f cycles per instruction (one wave per SIMD)\n", names[m], (double)h[m] / (steps * 16.0 * per[m]));
    return 0;
}
