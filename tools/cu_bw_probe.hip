// Microbenchmark: HBM-sourced bandwidth ONE compute unit reaches, as a function of how many CUs stream at
// the same time (grid = 1 ... 256 workgroups, one per CU: each allocates > half of the LDS) and of the
// transport: plain global_load_dwordx4 into registers, LDS-DMA (global_load_lds_dwordx4), global_store_dwordx4.
// Every workgroup walks its own contiguous cold region of a 4 GiB buffer (nothing is re-read).
//   hipcc -O3 --offload-arch=gfx950 tools/cu_bw_probe.hip -o tools/probe_bin/cu_bw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_HOLD = 96 * 1024;  // forces one workgroup per CU

// DEPTH loads of 16 B/lane in flight per wave, then consumed; rounds of them
template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void reg_stream(const char* src, size_t per_wg, int rounds, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* base = src + (size_t)blockIdx.x * per_wg + (size_t)wave * DEPTH * 1024 + lane * 16;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *(const u32x4*)(base + (size_t)r * WAVES * DEPTH * 1024 + d * 1024);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
    }
    if (acc[0] == 0x12345678u) sink[blockIdx.x] = acc[1] ^ acc[2] ^ acc[3] ^ (unsigned)(size_t)smem;
}

// LDS-DMA: DEPTH pieces of 1 KiB per wave per round, two rounds in flight
template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void dma_stream(const char* src, size_t per_wg, int rounds, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)blockIdx.x * per_wg + (size_t)wave * DEPTH * 1024 + lane * 16;
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)r * WAVES * DEPTH * 1024 + d * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + (((r & 1) * WAVES + wave) * DEPTH + d) * 1024), 16, 0, 0);
        if (r > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && ((unsigned*)smem)[lane] == 0x12345678u) sink[blockIdx.x] = 1;
}

template <int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void store_stream(char* dst, size_t per_wg, int rounds, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* base = dst + (size_t)blockIdx.x * per_wg + (size_t)wave * DEPTH * 1024 + lane * 16;
    const u32x4 v = {(unsigned)lane, 1u, 2u, (unsigned)(size_t)smem};
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) *(u32x4*)(base + (size_t)r * WAVES * DEPTH * 1024 + d * 1024) = v;
    }
}

// the MFMA-operand access pattern of the row-major residual stream: lane (li = lane&15, g = lane>>4) loads, for
// c = 0..11, 2 x 16 B at row li, float offset (g + 4c) * 8 of a [rows, 384] fp32 matrix (16-row fragments)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void frag_stream(const char* src, size_t per_wg, int rounds, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    // one round = one 16-row fragment per wave = 16 x 1536 B = 24 KiB
    const char* base = src + (size_t)blockIdx.x * per_wg + (size_t)wave * 24576 + (size_t)li * 1536 + g * 32;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[24];
        const char* b = base + (size_t)r * WAVES * 24576;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            v[2 * c] = *(const u32x4*)(b + c * 128);
            v[2 * c + 1] = *(const u32x4*)(b + c * 128 + 16);
        }
#pragma unroll
        for (int d = 0; d < 24; ++d) acc ^= v[d];
    }
    if (acc[0] == 0x12345678u) sink[blockIdx.x] = acc[1] ^ acc[2] ^ acc[3] ^ (unsigned)(size_t)smem;
}
// the same bytes in a fragment-blocked layout [rows/16][384/4][16 rows][4 floats]: the 16 lanes of a quarter-wave
// read 256 contiguous bytes
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void blocked_stream(const char* src, size_t per_wg, int rounds, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    const char* base = src + (size_t)blockIdx.x * per_wg + (size_t)wave * 24576 + (size_t)li * 16 + g * 512;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[24];
        const char* b = base + (size_t)r * WAVES * 24576;
#pragma unroll
        for (int c = 0; c < 12; ++c) {  // column quads 2(g + 4c), 2(g + 4c) + 1 -> 256-byte blocks
            v[2 * c] = *(const u32x4*)(b + c * 2048);
            v[2 * c + 1] = *(const u32x4*)(b + c * 2048 + 256);
        }
#pragma unroll
        for (int d = 0; d < 24; ++d) acc ^= v[d];
    }
    if (acc[0] == 0x12345678u) sink[blockIdx.x] = acc[1] ^ acc[2] ^ acc[3] ^ (unsigned)(size_t)smem;
}
// fragment-pattern stores (the epilogue): lane writes 16 B at row li, float offset 16 nf + 4 g, nf = 0..23
template <int WAVES, int BLOCKED>
__global__ __launch_bounds__(WAVES * 64) void frag_store(char* dst, size_t per_wg, int rounds, unsigned* sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, g = lane >> 4;
    char* base = dst + (size_t)blockIdx.x * per_wg + (size_t)wave * 24576 + (BLOCKED ? (size_t)li * 16 + g * 256 : (size_t)li * 1536 + g * 16);
    const u32x4 v = {(unsigned)lane, 1u, 2u, (unsigned)(size_t)smem};
    for (int r = 0; r < rounds; ++r) {
        char* b = base + (size_t)r * WAVES * 24576;
#pragma unroll
        for (int nf = 0; nf < 24; ++nf) *(u32x4*)(b + (BLOCKED ? nf * 1024 : nf * 64)) = v;
    }
}

template <typename K> void run(const char* name, K kern, int waves, int depth, char* buf, size_t cap, unsigned* sink) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_HOLD));
    printf("%-34s", name);
    for (int grid : {1, 8, 32, 64, 128, 256}) {
        const size_t per_wg = cap / 256;  // 16 MiB each
        const int rounds = (int)(per_wg / ((size_t)waves * depth * 1024));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), LDS_HOLD, 0, buf, per_wg, 8, sink);  // warm code, not data
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), LDS_HOLD, 0, buf + 0, per_wg, rounds, sink);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("  G=%3d %6.1f", grid, (double)per_wg / (ms * 1e-3) / 1e9);
        // evict: touch another 1 GiB so the next grid size does not find its lines in the Infinity Cache
        CK(hipMemsetAsync(buf + cap, 3, (size_t)1 << 30, 0));
    }
    printf("   GB/s per CU\n");
}

int main() {
    char* buf;
    unsigned* sink;
    const size_t cap = (size_t)4 << 30;
    CK(hipMalloc(&buf, cap + ((size_t)1 << 30)));
    CK(hipMemset(buf, 1, cap + ((size_t)1 << 30)));
    CK(hipMalloc(&sink, 4096));
    run("reg  4 waves x  8 x 16B/lane", reg_stream<4, 8>, 4, 8, buf, cap, sink);
    run("reg  4 waves x 24 x 16B/lane", reg_stream<4, 24>, 4, 24, buf, cap, sink);
    run("reg  4 waves x 48 x 16B/lane", reg_stream<4, 48>, 4, 48, buf, cap, sink);
    run("reg  8 waves x 24 x 16B/lane", reg_stream<8, 24>, 8, 24, buf, cap, sink);
    run("frag rows  4 waves x 24 x 16B", frag_stream<4>, 4, 24, buf, cap, sink);
    run("frag blocked 4 waves x 24 x 16B", blocked_stream<4>, 4, 24, buf, cap, sink);
    run("frag store rows   4 waves", frag_store<4, 0>, 4, 24, buf, cap, sink);
    run("frag store blocked 4 waves", frag_store<4, 1>, 4, 24, buf, cap, sink);
    run("dma  4 waves x  4 KiB (x2 rounds)", dma_stream<4, 4>, 4, 4, buf, cap, sink);
    run("dma  4 waves x  8 KiB (x2 rounds)", dma_stream<4, 8>, 4, 8, buf, cap, sink);
    run("dma  4 waves x 12 KiB (x2 rounds)", dma_stream<4, 12>, 4, 12, buf, cap, sink);
    run("store 4 waves x 8 x 16B/lane", store_stream<4, 8>, 4, 8, buf, cap, sink);
    run("store 4 waves x 24 x 16B/lane", store_stream<4, 24>, 4, 24, buf, cap, sink);
    return 0;
}
