// Issue-rate probe for v_exp_f32 vs v_fma_f32 on gfx950: N dependent-free instructions per wave, W waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/exp_rate.hip -o tools/bin/exp_rate && tools/bin/exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (MODE == 1)
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (MODE == 3)   // each exp reads the fma result produced by the instruction right before it
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_exp_f32 %1, %0\n v_fma_f32 %2, %2, %2, %2\n v_exp_f32 %3, %2\n v_fma_f32 %4, %4, %4, %4\n v_exp_f32 %5, %4\n v_fma_f32 %6, %6, %6, %6\n v_exp_f32 %7, %6"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else if (MODE == 4)   // exp result consumed by the next instruction
            asm volatile("v_exp_f32 %0, %0\n v_add_f32 %1, %0, %1\n v_exp_f32 %2, %2\n v_add_f32 %3, %2, %3\n v_exp_f32 %4, %4\n v_add_f32 %5, %4, %5\n v_exp_f32 %6, %6\n v_add_f32 %7, %6, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        else   // exp interleaved with fma (is the exp's cost issue or a separate unit?)
            asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_exp_f32 %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_exp_f32 %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_exp_f32 %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
void run(const char* name, int waves_per_simd) {
    float* out; hipMalloc(&out, 4 << 20);
    const int iters = 20000, blocks = 256 * waves_per_simd;   // 256 CUs x 4 SIMDs: one 256-thread block = 1 wave per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 8 * waves_per_simd;
    printf("%-28s waves/SIMD %d: %.1f ns per wave-instruction per SIMD (%.2f ms)\n", name, waves_per_simd, ms * 1e6 / instr_per_simd, ms);
    hipFree(out);
}
int main() {
    for (int w : {1, 2}) { run<0>("v_exp_f32", w); run<1>("v_fma_f32", w); run<2>("v_exp_f32 + v_fma_f32 pairs", w); run<3>("fma -> dependent exp pairs", w); run<4>("exp -> dependent add pairs", w); }
    return 0;
}
