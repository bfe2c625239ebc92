// Microbenchmark: how long a wave is held by the ISSUE of LDS-DMA pieces (1 KiB per instruction), from an L2-resident
// buffer, with empty queues: NP pieces back to back between two s_memtime reads, then a drain; 1 or 4 waves per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_issue_probe.hip -o tools/probe_bin/dma_issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int LDS = 144 * 1024;

template <int NP, int MODE>  // MODE 0 contiguous, 1 rows of 128 B (stride 768)
__global__ __launch_bounds__(256) void k(const char* src, int bytes, int iters, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t lo = MODE == 0 ? (size_t)lane * 16 : (size_t)(lane >> 3) * 768 + (lane & 7) * 16;
    unsigned long long issue = 0, total = 0;
    int pos = wave * NP;
    for (int it = 0; it < iters; ++it) {
        const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int d = 0; d < NP; ++d) {
            const size_t off = MODE == 0 ? (size_t)((pos + d) % (bytes / 1024)) * 1024 : (size_t)((pos + d) % (bytes / 6144)) * 6144;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off + lo),
                                             (__attribute__((address_space(3))) void*)(smem + (wave * NP + d) * 1024), 16, 0, 0);
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_readcyclecounter();
        issue += t1 - t0;
        total += t2 - t0;
        pos += 4 * NP;
        __builtin_amdgcn_s_sleep(20);
    }
    if (lane == 0) {
        out[(blockIdx.x * 4 + wave) * 2] = issue;
        out[(blockIdx.x * 4 + wave) * 2 + 1] = total;
    }
}

template <typename K> void run(const char* name, K kern, int np, int threads, char* buf, int bytes, unsigned long long* out) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), LDS, 0, buf, bytes, 50, out);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), LDS, 0, buf, bytes, iters, out);
    CK(hipDeviceSynchronize());
    static unsigned long long h[256 * 8];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    double is = 0, to = 0;
    const int nw = threads / 64;
    for (int b = 0; b < grid; ++b)
        for (int w = 0; w < nw; ++w) {
            is += (double)h[(b * 4 + w) * 2];
            to += (double)h[(b * 4 + w) * 2 + 1];
        }
    is /= (double)grid * nw * iters;
    to /= (double)grid * nw * iters;
    printf("%-40s %d waves/CU, %2d pieces: issue %.0f cyc (%.1f / piece), issue+landed %.0f cyc\n", name, nw, np, is, is / np, to);
}

int main() {
    const int bytes = 2359296;
    char* buf;
    unsigned long long* out;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&out, 256 * 8 * 8));
    CK(hipMemset(buf, 1, bytes));
    run("contiguous", k<2, 0>, 2, 64, buf, bytes, out);
    run("contiguous", k<2, 0>, 2, 256, buf, bytes, out);
    run("contiguous", k<4, 0>, 4, 256, buf, bytes, out);
    run("contiguous", k<12, 0>, 12, 64, buf, bytes, out);
    run("contiguous", k<12, 0>, 12, 256, buf, bytes, out);
    run("8 rows x 128 B", k<2, 1>, 2, 64, buf, bytes, out);
    run("8 rows x 128 B", k<2, 1>, 2, 256, buf, bytes, out);
    run("8 rows x 128 B", k<12, 1>, 12, 64, buf, bytes, out);
    run("8 rows x 128 B", k<12, 1>, 12, 256, buf, bytes, out);
    return 0;
}
