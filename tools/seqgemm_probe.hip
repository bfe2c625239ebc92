// Stand-alone harness for the A-stationary GEMM (hipt_abmil_atec23_amd/csrc/seqgemm.hip included as source):
// the ViT-256 QKV (LayerNorm fused, K 384 -> N 1152) and proj (K 384 -> N 384) shapes on random data.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I hipt_abmil_atec23_amd/csrc -I include tools/seqgemm_probe.hip -o tools/probe_bin/seqgemm_probe
//   run:   [HIPT_SEQGEMM_STAMPS=1] tools/probe_bin/seqgemm_probe [regions | check [M]] [qkv|proj]
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <vector>

#ifdef PROBE_OLD  // the one-shot kernel (seqgemm.hip)
#include "../hipt_abmil_atec23_amd/csrc/seqgemm.hip"
bool hipt_seqgemm_pipe_supported(int, int, int, bool, int) { return false; }
int hipt_seqgemm_pipe_launch(const SeqGemmParams&, bool, hipStream_t) { return -1; }
#define LAUNCH(p, ln) hipt_seqgemm_launch(p, ln, 0, 0)
#else  // the pipelined kernel; "dbg N" as last-but-one arguments selects a debug build (1 no DMA, 2 no stores, 4 no MFMA)
#include "../hipt_abmil_atec23_amd/csrc/seqgemm_pipe.hip"
static int g_dbg = 0;
static int launch_dbg(const SeqGemmParams& p, bool ln) {
    switch (g_dbg * 2 + (ln ? 1 : 0)) {
        case 1: return hipt_seqgemm_pipe_launch_dbg<true, 0>(p, 0);
        case 0: return hipt_seqgemm_pipe_launch_dbg<false, 0>(p, 0);
        case 3: return hipt_seqgemm_pipe_launch_dbg<true, 1>(p, 0);
        case 5: return hipt_seqgemm_pipe_launch_dbg<true, 2>(p, 0);
        case 7: return hipt_seqgemm_pipe_launch_dbg<true, 3>(p, 0);
        case 9: return hipt_seqgemm_pipe_launch_dbg<true, 4>(p, 0);
        default: return -1;
    }
}
#define LAUNCH(p, ln) launch_dbg(p, ln)
#endif

void hipt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static unsigned lcg = 12345u;
static float frand() {
    lcg = lcg * 1664525u + 1013904223u;
    return ((lcg >> 8) & 0xffff) / 32768.0f - 1.0f;
}
static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16);
}
static float bf2f(uint16_t b) {
    uint32_t u = (uint32_t)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv) {
    const bool check = argc > 1 && strcmp(argv[1], "check") == 0;
    const int regions = (argc > 1 && !check) ? atoi(argv[1]) : 1;
    const int M = check ? (argc > 2 && atoi(argv[2]) > 0 ? atoi(argv[2]) : 514) : regions * 65792, K = 384;
    const bool proj = strcmp(argv[argc - 1], "proj") == 0;
    const int N = proj ? 384 : 1152;
    std::vector<float> hx((size_t)M * K), hb(N), hg(K), hbt(K);
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    for (auto& v : hx) v = frand();
    for (size_t i = 0; i < ha.size(); ++i) ha[i] = f2bf(hx[i]);
    for (auto& v : hw) v = f2bf(frand() * 0.05f);
    for (auto& v : hb) v = frand() * 0.1f;
    for (auto& v : hg) v = 1.0f + frand() * 0.1f;
    for (auto& v : hbt) v = frand() * 0.1f;
    void *x, *a, *w, *b, *g, *bt, *out, *ctr;
    CK(hipMalloc(&ctr, 64));
    CK(hipMalloc(&x, hx.size() * 4));
    CK(hipMalloc(&a, ha.size() * 2));
    CK(hipMalloc(&w, hw.size() * 2));
    CK(hipMalloc(&b, N * 4));
    CK(hipMalloc(&g, K * 4));
    CK(hipMalloc(&bt, K * 4));
    CK(hipMalloc(&out, (size_t)M * N * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bt, hbt.data(), K * 4, hipMemcpyHostToDevice));
    SeqGemmParams p{};
    p.A = proj ? a : x; p.lda = K; p.ln_w = (float*)g; p.ln_b = (float*)bt; p.ln_eps = 1e-6f; p.W = w; p.M = M; p.N = N; p.K = K;
    p.bias = (float*)b; p.out = out; p.ldc = N; p.counter = (int*)ctr;
#ifndef PROBE_OLD
    if (const char* e = getenv("PROBE_DBG")) g_dbg = atoi(e);
#endif
    if (check) {
        if (LAUNCH(p, !proj)) return 2;
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> ho((size_t)M * N);
        CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        double maxerr = 0;
        long nbad = 0;
        int br = -1, bc = -1;
        std::vector<double> av(K);
        for (int r = 0; r < M; ++r) {
            if (proj) {
                for (int k = 0; k < K; ++k) av[k] = bf2f(ha[(size_t)r * K + k]);
            } else {
                double mean = 0, var = 0;
                for (int k = 0; k < K; ++k) mean += hx[(size_t)r * K + k];
                mean /= K;
                for (int k = 0; k < K; ++k) var += (hx[(size_t)r * K + k] - mean) * (hx[(size_t)r * K + k] - mean);
                const double rstd = 1.0 / sqrt(var / K + 1e-6);
                for (int k = 0; k < K; ++k) av[k] = bf2f(f2bf((float)((hx[(size_t)r * K + k] - mean) * rstd * hg[k] + hbt[k])));
            }
            for (int n = 0; n < N; ++n) {
                double acc = hb[n];
                for (int k = 0; k < K; ++k) acc += av[k] * bf2f(hw[(size_t)n * K + k]);
                const double got = bf2f(ho[(size_t)r * N + n]);
                const double e = fabs(got - acc);
                if (!(e <= 1e30)) { ++nbad; if (br < 0) { br = r; bc = n; } continue; }
                if (e > maxerr) maxerr = e;
                if (e > 2e-2 + 8e-3 * fabs(acc)) { ++nbad; if (br < 0) { br = r; bc = n; } }
            }
        }
        {   // batch invariance: the same rows at a different position inside the tiles must give the same bits
            const int sh = argc > 4 ? atoi(argv[3]) : 12;
            SeqGemmParams p2 = p;
            p2.A = proj ? (const void*)((const uint16_t*)a + (size_t)sh * K) : (const void*)((const float*)x + (size_t)sh * K);
            p2.M = M - sh;
            std::vector<uint16_t> ho2((size_t)M * N);
            CK(hipMemset(out, 0, (size_t)M * N * 2));
            if (LAUNCH(p2, !proj)) return 2;
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ho2.data(), out, ho2.size() * 2, hipMemcpyDeviceToHost));
            long ndiff = 0;
            int fr = -1, fc = -1;
            for (int r = sh; r < M; ++r)
                for (int n = 0; n < N; ++n)
                    if (ho[(size_t)r * N + n] != ho2[(size_t)(r - sh) * N + n]) {
                        ++ndiff;
                        if (fr < 0) { fr = r; fc = n; }
                    }
            printf("shift-by-%d invariance: %ld elements differ bitwise, first (row %d, col %d)\n", sh, ndiff, fr, fc);
            nbad += ndiff;
        }
        printf("check %s M=%d: max |err| %.3e, %ld elements off, first bad (row %d, col %d)\n", proj ? "proj" : "qkv", M, maxerr, nbad, br, bc);
        return nbad ? 1 : 0;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) LAUNCH(p, !proj);
    CK(hipDeviceSynchronize());
    const int iters = 5;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) LAUNCH(p, !proj);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("%s M=%d N=%d: %.1f us/launch  %.1f TFLOP/s  %.2f TB/s (algorithmic bytes)\n", proj ? "proj" : "qkv", M, N, us, 2.0 * M * N * K / us / 1e6,
           ((double)M * K * (proj ? 2 : 4) + (double)M * N * 2) / us / 1e6);
    return 0;
}
