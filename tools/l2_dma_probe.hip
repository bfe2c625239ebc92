// Microbenchmark: L2 -> LDS bandwidth per CU when EVERY workgroup streams the SAME small buffer (the weight stream of
// the fused MLP: 2.36 MB per 128-row tile, re-read by all 256 workgroups), as a function of the request shape:
//   contig : one 1 KiB piece = 64 lanes x 16 B contiguous
//   rows   : one piece = 8 rows x 128 B, rows `stride` bytes apart (the swizzled operand image of a [*, K] bf16 matrix)
// and of the number of pieces in flight per wave.
//   hipcc -O3 --offload-arch=gfx950 tools/l2_dma_probe.hip -o tools/probe_bin/l2_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int LDS = 144 * 1024;

// MODE 0: contiguous pieces; MODE 1: 8 rows x 128 B (stride bytes apart); MODE 2: as 1 but plain loads to registers
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream(const char* src, int bytes, int stride, int passes, int rot, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npiece = bytes / 1024;  // 1 KiB pieces; wave w takes pieces w, w+4, ...
    const int per_row = stride / 128;  // 128-B segments per row
    unsigned acc = 0;
    int slot = 0;
    const int start = rot ? (blockIdx.x * 37) % npiece : 0;
    for (int p = 0; p < passes; ++p) {
        for (int i0 = wave; i0 < npiece; i0 += 4 * DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                int i = i0 + 4 * d + start;
                i = i >= npiece ? i - npiece : i;
                i = i >= npiece ? i - npiece : i;
                size_t off;
                if (MODE == 0) {
                    off = (size_t)i * 1024 + lane * 16;
                } else {
                    // piece i = rows 8*(i / per_row) .. +7, segment i % per_row
                    const int r = 8 * (i / per_row) + (lane >> 3), seg = i % per_row;
                    off = (size_t)r * stride + seg * 128 + (lane & 7) * 16;
                }
                if (MODE == 2) {
                    const uint4 v = *(const uint4*)(src + off);
                    acc ^= v.x ^ v.y ^ v.z ^ v.w;
                } else {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off),
                                                     (__attribute__((address_space(3))) void*)(smem + ((slot * 4 + wave) * DEPTH + d) * 1024), 16, 0, 0);
                }
            }
            slot = slot == 2 ? 0 : slot + 1;
            if (MODE != 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DEPTH) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && (((unsigned*)smem)[lane] ^ acc) == 0x12345678u) sink[blockIdx.x] = 1;
}

template <typename K> void run(const char* name, K kern, char* buf, int bytes, int stride, int rot, unsigned* sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    printf("%-44s", name);
    for (int grid : {1, 32, 256}) {
        const int passes = 40;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, 0, buf, bytes, stride, 2, rot, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, 0, buf, bytes, stride, passes, rot, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("  G=%3d %6.1f", grid, (double)bytes * passes / (ms * 1e-3) / 1e9);
    }
    printf("   GB/s per CU\n");
}

int main() {
    const int bytes = 2359296;  // W1 + W2 of ViT-256 in bf16
    char* buf;
    unsigned* sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(buf, 1, bytes));
    run("DMA contiguous 1 KiB, 4 in flight/wave", stream<0, 2>, buf, bytes, 768, 0, sink);
    run("DMA contiguous 1 KiB, 8 in flight/wave", stream<0, 4>, buf, bytes, 768, 0, sink);
    run("DMA contiguous 1 KiB, 16 in flight/wave", stream<0, 8>, buf, bytes, 768, 0, sink);
    run("DMA 8 rows x 128 B stride 768, 8 in flight", stream<1, 4>, buf, bytes, 768, 0, sink);
    run("DMA 8 rows x 128 B stride 3072, 8 in flight", stream<1, 4>, buf, bytes, 3072, 0, sink);
    run("DMA 8 rows x 128 B stride 768, 16 in flight", stream<1, 8>, buf, bytes, 768, 0, sink);
    run("DMA contiguous, 8 in flight, rotated start", stream<0, 4>, buf, bytes, 768, 1, sink);
    run("DMA rows 768, 8 in flight, rotated start", stream<1, 4>, buf, bytes, 768, 1, sink);
    run("plain loads rows 768, 8 in flight", stream<2, 4>, buf, bytes, 768, 0, sink);
    run("plain loads rows 768, 16 in flight", stream<2, 8>, buf, bytes, 768, 0, sink);
    return 0;
}
