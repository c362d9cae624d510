// FETCH_SIZE calibration for gfx950: three kernels that each read a buffer of KNOWN size exactly once
// with one access pattern, to be run under `rocprofv3 --pmc FETCH_SIZE` (tools/fetch_calib.py does that and prints
// counter / bytes).  The MI355X guide says FETCH_SIZE reports 1/2 of the bytes of a wide coalesced stream; the question is
// whether that also holds for the two LDS-DMA shapes the engine's dominant kernels use:
//   ldsdma_linear   one global_load_lds_dwordx4 = 64 lanes x 16 B contiguous (1 KB)        -- mlp_fused.hip weight stream
//   ldsdma_8x128    one global_load_lds_dwordx4 = 8 rows x 128 B, row stride 1 KB, 16-byte chunks XOR-swizzled per row
//                                                                                           -- gemm.hip stage() (A operand, K = 512)
//   vgpr_dwordx4    plain coalesced global_load_dwordx4 into registers (the guide's reference case)
// Build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kBlocks = 2048, kThreads = 256;

__global__ void __launch_bounds__(256) ldsdma_linear(const char* src, size_t bytes, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[4][8][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per_wave = bytes / ((size_t)kBlocks * 4);
    const char* p = src + ((size_t)blockIdx.x * 4 + wave) * per_wave + lane * 16;
    for (size_t off = 0; off < per_wave; off += 8 * 1024) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(p + off + j * 1024), (lptr_t)&lds[wave][j][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (sink && lane == 0) sink[blockIdx.x * 4 + wave] = *(volatile unsigned*)&lds[wave][0][0];
}

__global__ void __launch_bounds__(256) ldsdma_8x128(const char* src, size_t bytes, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char lds[4][8][1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr size_t kRow = 1024;                                  // bytes per matrix row (K = 512 bf16)
    const size_t rows = bytes / kRow, rows_per_wave = rows / ((size_t)kBlocks * 4);
    const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * rows_per_wave;
    const int lr = lane >> 3;
    for (size_t r = 0; r < rows_per_wave; r += 8) {                // 8 rows x 8 k-tiles of 128 B = 8 KB per trip
        const size_t row = r0 + r + lr;
        const int chunk = (lane & 7) ^ (((int)row >> 1) & 7);      // source-side swizzle of gemm.hip
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + row * kRow + kt * 128 + chunk * 16), (lptr_t)&lds[wave][kt][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (sink && lane == 0) sink[blockIdx.x * 4 + wave] = *(volatile unsigned*)&lds[wave][0][0];
}

__global__ void __launch_bounds__(256) vgpr_dwordx4(const char* src, size_t bytes, unsigned* sink) {
    const size_t n16 = bytes / 16, stride = (size_t)kBlocks * kThreads;
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n16; i += stride) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

__global__ void fill(unsigned* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = (unsigned)i * 2654435761u ^ seed;
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 1024) << 20;     // MiB; default 1 GiB (4x the Infinity Cache)
    char *buf = nullptr, *evict = nullptr;
    unsigned* sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&evict, (size_t)768 << 20));
    CHECK(hipMalloc(&sink, kBlocks * 4 * sizeof(unsigned)));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (unsigned*)buf, bytes / 4, 1u);
    auto flush = [&]() { hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (unsigned*)evict, ((size_t)768 << 20) / 4, 2u); };
    flush();
    hipLaunchKernelGGL(ldsdma_linear, dim3(kBlocks), dim3(kThreads), 0, 0, buf, bytes, sink);
    flush();
    hipLaunchKernelGGL(ldsdma_8x128, dim3(kBlocks), dim3(kThreads), 0, 0, buf, bytes, sink);
    flush();
    hipLaunchKernelGGL(vgpr_dwordx4, dim3(kBlocks), dim3(kThreads), 0, 0, buf, bytes, sink);
    CHECK(hipDeviceSynchronize());
    printf("bytes_per_kernel %zu\n", bytes);
    return 0;
}
