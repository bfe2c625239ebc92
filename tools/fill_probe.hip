// Microbenchmark: how fast can one CU pull L2-resident slabs into LDS by LDS-DMA (global_load_lds_dwordx4)
// and into registers (global_load_dwordx4)?  Calibrates the GEMM tiling (bytes of LDS fill per MFMA flop).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int DEPTH>  // each wave issues DEPTH x 1 KiB DMA pieces per round, waits for the oldest round
__global__ __launch_bounds__(WAVES * 64) void lds_fill(const char* src, size_t span, int rounds, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // every workgroup walks the same `span` bytes (L2-resident), offset by block to decorrelate
    size_t off = ((size_t)blockIdx.x * 7919 * 1024) % span;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const char* g = src + (off + ((size_t)(wave * DEPTH + d) * 1024 + lane * 16)) % span;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(smem + ((r & 1) * WAVES * DEPTH + wave * DEPTH + d) * 1024), 16, 0, 0);
        }
        off = (off + (size_t)WAVES * DEPTH * 1024) % span;
        if (r > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");  // previous round landed
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((unsigned*)smem)[lane];
}

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void reg_fill(const char* src, size_t span, int rounds, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off = ((size_t)blockIdx.x * 7919 * 1024) % span;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            v[d] = *(const u32x4*)(src + (off + ((size_t)(wave * DEPTH + d) * 1024 + lane * 16)) % span);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        off = (off + (size_t)WAVES * DEPTH * 1024) % span;
    }
    if (acc[0] == 0x12345678u) sink[blockIdx.x] = acc[1] ^ acc[2] ^ acc[3];
}

template <typename K> void run(const char* name, K kern, int waves, int depth, const char* src, size_t span, unsigned* sink, int lds) {
    const int rounds = 2000, grid = 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    if (lds > 65536) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, src, span, 50, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), lds, 0, src, span, rounds, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)grid * rounds * waves * depth * 1024.0;
    printf("%-28s span %6.1f MB: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, span / 1e6, bytes / grid / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    char* src; unsigned* sink;
    const size_t cap = 512u << 20;
    CK(hipMalloc(&src, cap)); CK(hipMemset(src, 1, cap)); CK(hipMalloc(&sink, 4096));
    for (size_t span : {(size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20, cap}) {
        run("lds_dma 8 waves x 4 KiB", lds_fill<8, 4>, 8, 4, src, span, sink, 2 * 8 * 4 * 1024);
        run("lds_dma 8 waves x 8 KiB", lds_fill<8, 8>, 8, 8, src, span, sink, 2 * 8 * 8 * 1024);
        run("lds_dma 4 waves x 8 KiB", lds_fill<4, 8>, 4, 8, src, span, sink, 2 * 4 * 8 * 1024);
        run("reg 8 waves x 8 x16B", reg_fill<8, 8>, 8, 8, src, span, sink, 0);
        run("reg 16 waves x 8 x16B", reg_fill<16, 8>, 16, 8, src, span, sink, 0);
    }
    return 0;
}
