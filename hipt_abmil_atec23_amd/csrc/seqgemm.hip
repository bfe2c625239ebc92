// A-STATIONARY GEMM for the K <= 384 projections of a transformer block (bf16 mode):
//   out = epilogue( LN?(x)[rows, K] @ W[N,K]^T + bias )
// used for QKV (LayerNorm-1 fused), attn.proj (+residual) and mlp.fc1 (LayerNorm-2 fused, GELU)
// (HIPT_4K/vision_transformer.py:93-94,99-100,114,116,121,129,147,151).
//
// Why: on gfx950 a CU can pull ~70 GB/s from its XCD's L2 into LDS (tools/fill_probe.hip), i.e. an
// LDS-tiled GEMM needs >= 140 flop per byte of LDS fill to be MFMA-bound; a 272x128 tile re-streaming
// its A slab for every N tile has 85 and was measured at 20 % MFMA busy (60 % of wave cycles waiting).
// Here ONE 4-wave workgroup owns 256 rows (16 MFMA row fragments, 4 per wave) and keeps the whole
// [256, K] activation tile as MFMA operand fragments IN REGISTERS (4 x K/8 chunks x 4 = 192 VGPRs at
// one wave per SIMD).  Only the weights move: 16 KiB slabs (128 output columns x 64 k) through an
// 8-slot LDS ring filled by LDS-DMA, one raw s_barrier per slab, counted vmcnt.  Every W fragment read
// from LDS feeds 4 MFMAs per wave; activations are read from HBM exactly once.
// LayerNorm is fused into the activation load: the 4 lanes (lane>>4 = 0..3) that share a row hold the
// complete row, so mean / centred variance are two xor-shuffles — the LN kernels and their [M, D]
// round trip disappear.
// Tiling: these ops are row-wise, so the flat M = 256 x 257 rows are cut into 256-row tiles regardless
// of sequence boundaries: 257 tiles.  The tiles of the last partial round (1 here) are split across
// their N tiles into separate small workgroups, so the tail costs ~1/tiles_n of a round, not a round.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int TM = 256;                // rows per workgroup
constexpr int SLAB_BYTES = 128 * 128;  // 128 W rows x 64 bf16
constexpr int NSLOT = 8;
constexpr int GB_BYTES = 2 * 384 * 4;  // gamma | beta staging (K <= 384)
constexpr int MAXN = 2048;             // bias staging (floats)
constexpr int SEQ_LDS = NSLOT * SLAB_BYTES + GB_BYTES + MAXN * 4;

template <int N> __device__ __forceinline__ void wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `slabs` later slabs (4 DMA instructions each, per wave) are still in flight
__device__ __forceinline__ void wait_slabs(int slabs) {
    switch (slabs) {
        case 0: wait_vm_n<0>(); break;
        case 1: wait_vm_n<4>(); break;
        case 2: wait_vm_n<8>(); break;
        case 3: wait_vm_n<12>(); break;
        case 4: wait_vm_n<16>(); break;
        case 5: wait_vm_n<20>(); break;
        default: wait_vm_n<24>(); break;
    }
}

// Epilogue of 4 consecutive columns of one row.  No global LOADS here on purpose: with LDS-DMA in
// flight hipcc waits vmcnt(0) before using any ordinary load result, which would drain every
// outstanding store and the whole W ring once per fragment (measured: 2-3x kernel time).  The bias comes
// from LDS; the residual add is deferred to the consumer kernel's activation load.
template <int FLAGS>
__device__ __forceinline__ void seq_epilogue(const SeqGemmParams& p, uint32_t bias_lds, int64_t row, int n, f32x4 v) {
    v += lds_ld128(bias_lds + n * 4);  // asm LDS read: a visible one would cost a vmcnt(0) ring drain (common.h)
    if constexpr (FLAGS & HIPT_EPI_GELU) {
        {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
        }
    }
    store4<bf16_t>((bf16_t*)p.out + row * p.ldc + n, v);
}

// LayerNorm of one row held by the 4 lanes (lane>>4 = 0..3) that share lane&15: lane owns chunks
// g + 4c (8 elements each).  Two-pass statistics (mean, centred variance) as torch; output bf16 fragments.
template <int NCH>
__device__ __forceinline__ void ln_rows(f32x4 (&v)[NCH][2], const float* gam, const float* bet, int K, float eps, int g,
                                        u32x4 (&out)[NCH]) {
#pragma clang fp contract(off)  // every multiply-add written out: all unrolled instances must round alike (pipe_common.h)
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[c][0][e] + v[c][1][e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)K;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = v[c][0][e] - mean, b = v[c][1][e] - mean;
            q = __builtin_fmaf(a, a, q);
            q = __builtin_fmaf(b, b, q);
        }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q / (float)K + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = (g + 4 * c) * 8;
        const f32x4 g0 = *(const f32x4*)(gam + k0), g1 = *(const f32x4*)(gam + k0 + 4);
        const f32x4 b0 = *(const f32x4*)(bet + k0), b1 = *(const f32x4*)(bet + k0 + 4);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y0[e] = __builtin_fmaf((v[c][0][e] - mean) * rstd, g0[e], b0[e]);
            y1[e] = __builtin_fmaf((v[c][1][e] - mean) * rstd, g1[e], b1[e]);
        }
        u32x4 o;
        o[0] = pack_bf16x2(y0[0], y0[1]);
        o[1] = pack_bf16x2(y0[2], y0[3]);
        o[2] = pack_bf16x2(y1[0], y1[1]);
        o[3] = pack_bf16x2(y1[2], y1[3]);
        out[c] = o;
    }
}

#define STAMP(k)                                                                                   \
    do {                                                                                           \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

template <int KS, bool LN, int FLAGS>
__global__ __launch_bounds__(256, 1) void seqgemm_kernel(const SeqGemmParams p) {
    constexpr int K = KS * 64;
    constexpr int NCH = KS * 2;  // 16-byte A chunks per lane per fragment (one per (slab, ks))
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gam = (float*)(smem + NSLOT * SLAB_BYTES);
    float* bet = gam + 384;
    float* bia = bet + 384;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    // blocks [0, full_tiles): one whole row tile each; then tail tiles split nsplit ways over N tiles
    int tile, split, nsplit;
    if ((int)blockIdx.x < p.full_tiles) {
        tile = blockIdx.x;
        split = 0;
        nsplit = 1;
    } else {
        const int b = blockIdx.x - p.full_tiles;
        tile = p.full_tiles + b / p.nsplit;
        split = b % p.nsplit;
        nsplit = p.nsplit;
    }
    const int row0 = tile * TM;
    int nrows = p.M - row0;
    nrows = nrows < TM ? nrows : TM;
    const int tiles_n = (p.N + 127) >> 7;
    const int my_tiles = (tiles_n - split + nsplit - 1) / nsplit;  // N tiles split, split+nsplit, ...
    const int nslab = my_tiles * KS;
    // De-phase the workgroups: each starts at a different N tile and wraps.  In lockstep every store of
    // the chip would land in the same 256-byte column window at a row stride of ldc -> a quarter of the
    // HBM channels (measured 1.7 TB/s); rotated, the windows cover all columns at any time.
    const int rot = tile % my_tiles;
    auto ntile_of = [&](int t) {
        int u = t + rot;
        u = u >= my_tiles ? u - my_tiles : u;
        return split + u * nsplit;
    };

    // ---- W slab DMA: slab i = (N tile split + (i / KS) * nsplit, k slab i % KS); 4 instructions per wave ----
    const bf16_t* W = (const bf16_t*)p.W;
    auto issue = [&](int i) {
        const int nt = ntile_of(i / KS), kt = i % KS;
        char* dst = smem + (i % NSLOT) * SLAB_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int blk = q * 4 + wave;
            const int r = blk * 8 + (lane >> 3);
            int n = nt * 128 + r;
            n = n < p.N ? n : p.N - 1;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            glds16(W + (int64_t)n * K + (kt * 8 + c) * 8, dst + blk * 1024);
        }
    };
    STAMP(0);
    const int pre = nslab < NSLOT - 1 ? nslab : NSLOT - 1;
    for (int i = 0; i < pre; ++i) issue(i);

    // ---- activations -> registers (B' operand fragments), LayerNorm fused ----
    for (int i = tid; i < p.N; i += 256) bia[i] = p.bias ? p.bias[i] : 0.f;
    if constexpr (LN) {
        for (int i = tid; i < K; i += 256) {
            gam[i] = p.ln_w[i];
            bet[i] = p.ln_b[i];
        }
    }
    __syncthreads();  // NB: also drains the W prefetch (vmcnt(0)); one-off
    STAMP(1);
    u32x4 af[4][NCH];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        int r = (wave * 4 + mf) * 16 + li;
        r = r < nrows ? r : nrows - 1;  // rows past the end re-read the last valid row (never stored)
        if constexpr (LN) {
            const float* xr = (const float*)p.A + (int64_t)(row0 + r) * p.lda;
            f32x4 v[NCH][2];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                v[c][0] = *(const f32x4*)(xr + (g + 4 * c) * 8);
                v[c][1] = *(const f32x4*)(xr + (g + 4 * c) * 8 + 4);
            }
            ln_rows<NCH>(v, gam, bet, K, p.ln_eps, g, af[mf]);
        } else {
            const bf16_t* ar = (const bf16_t*)p.A + (int64_t)(row0 + r) * p.lda;
#pragma unroll
            for (int c = 0; c < NCH; ++c) af[mf][c] = *(const u32x4*)(ar + (g + 4 * c) * 8);
        }
    }
    STAMP(2);

    // ---- fragment read offsets inside a W slab: row li of column fragment, chunk (g + 4ks) swizzled ----
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = li * 128 + (((g + 4 * ks) ^ ((lane >> 1) & 7)) << 4);

    const uint32_t lbase = (uint32_t)(uintptr_t)(LDS_AS char*)smem;  // LDS byte address of the ring
    f32x4 acc[4][8];
    for (int t = 0; t < my_tiles; ++t) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KS; ++kt) {
            const int i = t * KS + kt;
            // slab i has landed when at most the later slabs already issued remain in flight
            const int issued_after = (i + NSLOT - 2 < nslab ? i + NSLOT - 2 : nslab - 1) - i;
            wait_slabs(issued_after);
            __builtin_amdgcn_s_barrier();  // slab i visible to all; all waves are done reading slab i-1
            if (i + NSLOT - 1 < nslab) issue(i + NSLOT - 1);  // reuses the slot of slab i-1
            // 8 groups per slab: (ks, column-fragment pair), 8 MFMAs each.  Two explicit W register sets:
            // while the MFMAs of one group run, the NEXT group's pair is already in flight from LDS
            // (inline-asm ds_read_b128 + counted lgkmcnt; hipcc otherwise sinks the reads below the MFMAs
            // to reuse the registers and waits lgkmcnt(0) in front of every group).
            const uint32_t a0 = lbase + (i % NSLOT) * SLAB_BYTES + foff[0];
            const uint32_t a1 = lbase + (i % NSLOT) * SLAB_BYTES + foff[1];
            u32x4 wa0, wa1, wb0, wb1;
#define DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define LGKM(n)                                          \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define MMA8(w0, w1, ks, pr)                                                 \
    _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                       \
        Tr<bf16_t>::mma16(acc[mf][2 * (pr)], w0, af[mf][kt * 2 + (ks)]);     \
        Tr<bf16_t>::mma16(acc[mf][2 * (pr) + 1], w1, af[mf][kt * 2 + (ks)]); \
    }                                                                        \
    __builtin_amdgcn_sched_barrier(0)
            DSR(wa0, a0, 0); DSR(wa1, a0, 2048);
            DSR(wb0, a0, 4096); DSR(wb1, a0, 6144);
            LGKM(2); MMA8(wa0, wa1, 0, 0);
            DSR(wa0, a0, 8192); DSR(wa1, a0, 10240);
            LGKM(2); MMA8(wb0, wb1, 0, 1);
            DSR(wb0, a0, 12288); DSR(wb1, a0, 14336);
            LGKM(2); MMA8(wa0, wa1, 0, 2);
            DSR(wa0, a1, 0); DSR(wa1, a1, 2048);
            LGKM(2); MMA8(wb0, wb1, 0, 3);
            DSR(wb0, a1, 4096); DSR(wb1, a1, 6144);
            LGKM(2); MMA8(wa0, wa1, 1, 0);
            DSR(wa0, a1, 8192); DSR(wa1, a1, 10240);
            LGKM(2); MMA8(wb0, wb1, 1, 1);
            DSR(wb0, a1, 12288); DSR(wb1, a1, 14336);
            LGKM(2); MMA8(wa0, wa1, 1, 2);
            LGKM(0); MMA8(wb0, wb1, 1, 3);
#undef DSR
#undef LGKM
#undef MMA8
        }
        if (t == 0) STAMP(3);
        // ---- epilogue of this 128-column tile: lane holds C[row = frag*16 + li][n0 + nf*16 + 4g + 0..3] ----
        const int n0 = ntile_of(t) * 128;
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int r = (wave * 4 + mf) * 16 + li;
            if (r < nrows) {
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) {
                    const int n = n0 + nf * 16 + 4 * g;
                    if (n < p.N) seq_epilogue<FLAGS>(p, lds_addr(bia), row0 + r, n, acc[mf][nf]);
                }
            }
        }
        if (t == 0) STAMP(4);
    }
    STAMP(5);
}

template <int KS, bool LN, int FLAGS>
int launch(const SeqGemmParams& p, int grid, hipStream_t st) {
    auto k = seqgemm_kernel<KS, LN, FLAGS>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, SEQ_LDS) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(seqgemm) failed");
            return HIPT_E_LAUNCH;
        }
        once.done[dev] = true;
    }
    SeqGemmParams q = p;
#ifdef HIPT_DEBUG_STAMPS  // diagnostic builds only (make DEBUG_STAMPS=1): the release library never allocates or synchronises
    static const bool want_stamps = getenv("HIPT_SEQGEMM_STAMPS") != nullptr;
    static unsigned long long* dbuf = nullptr;
    if (want_stamps) {
        if (!dbuf) (void)hipMalloc(&dbuf, 4096 * 8 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbuf, 0, 4096 * 8 * sizeof(unsigned long long), st);
        q.stamps = dbuf;
    }
#endif
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), SEQ_LDS, st, q);
    HIPT_CHECK_LAUNCH();
#ifdef HIPT_DEBUG_STAMPS
    if (want_stamps && grid <= 4096) {  // debug only: synchronises and prints phase medians (us)
        static unsigned long long h[4096 * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbuf, (size_t)grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t5 = 0;
        for (int b = 0; b < grid; ++b) {
            if (h[b * 8] < t0) t0 = h[b * 8];
            if (h[b * 8 + 5] > t5) t5 = h[b * 8 + 5];
        }
        double ph[5] = {0, 0, 0, 0, 0};
        double start_max = 0, end_med = 0;
        const int nb = grid < 256 ? grid : 256;
        for (int b = 0; b < nb; ++b) {
            for (int k2 = 0; k2 < 5; ++k2) ph[k2] += (double)(h[b * 8 + k2 + 1] - h[b * 8 + k2]) * 0.01 / nb;
            const double s0 = (double)(h[b * 8] - t0) * 0.01;
            if (s0 > start_max) start_max = s0;
            end_med += (double)(h[b * 8 + 5] - t0) * 0.01 / nb;
        }
        fprintf(stderr, "[seqgemm KS=%d LN=%d F=%d N=%d grid=%d] total %.1f us | first-256 WGs: start<=%.1f, stage %.1f, Aload %.1f, tile0-k %.1f, tile0-epi %.1f, rest %.1f, end(avg) %.1f\n",
                KS, (int)LN, FLAGS, p.N, grid, (double)(t5 - t0) * 0.01, start_max, ph[0], ph[1], ph[2], ph[3], ph[4], end_med);
    }
#endif
    return HIPT_OK;
}

template <int KS>
int dispatch(const SeqGemmParams& p, int grid, bool ln, int flags, hipStream_t st) {
    if (ln && flags == 0) return launch<KS, true, 0>(p, grid, st);                          // LN1 + qkv
    if (ln && flags == HIPT_EPI_GELU) return launch<KS, true, HIPT_EPI_GELU>(p, grid, st);  // LN2 + fc1 + GELU
    if (!ln && flags == 0) return launch<KS, false, 0>(p, grid, st);                        // proj (branch output, bf16)
    hipt_set_error("seqgemm: unsupported variant ln=%d flags=%d", (int)ln, flags);
    return HIPT_E_UNSUPPORTED;
}

}  // namespace

bool hipt_seqgemm_supported(int dtype, int K) { return dtype == HIPT_BF16 && (K == 384 || K == 192); }

int hipt_seqgemm_launch(const SeqGemmParams& p_in, bool ln, int flags, hipStream_t st) {
    SeqGemmParams p = p_in;
    HIPT_CHECK_ARG(p.M > 0 && p.N > 0 && p.N % 4 == 0 && p.N <= MAXN, "seqgemm: bad shape M=%d N=%d", p.M, p.N);
    HIPT_CHECK_ARG(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.out % 16) == 0 && p.ldc % 4 == 0 &&
                       (p.lda * (ln ? 4 : 2)) % 16 == 0,
                   "seqgemm: 16-byte alignment required");
    // the pipelined kernel streams a pre-packed weight image: callers without one (NULL *_pk) get the generic kernel
    if (!hipt_generic_only() && p.counter && p.wpk && hipt_seqgemm_pipe_supported(HIPT_BF16, p.K, p.N, ln, flags)) return hipt_seqgemm_pipe_launch(p, ln, st);
    HIPT_CHECK_ARG(p.img == 0, "seqgemm: activation images / head-major output exist only in the pipelined kernel (img=%d)", p.img);
    const int tiles_n = (p.N + 127) / 128, tiles_m = (p.M + TM - 1) / TM;
    // whole rounds of 256 CUs run one workgroup per row tile; the tiles of the last partial round are
    // split over their N tiles when that round would otherwise be mostly empty
    const int rem = tiles_m % 256;
    const int tail = (rem > 0 && rem * tiles_n <= 512) ? rem : 0;
    p.full_tiles = tiles_m - tail;
    p.nsplit = tiles_n;
    const int grid = p.full_tiles + tail * p.nsplit;
    if (p.K == 384) return dispatch<6>(p, grid, ln, flags, st);
    if (p.K == 192) return dispatch<3>(p, grid, ln, flags, st);
    hipt_set_error("seqgemm: K=%d not in {192, 384}", p.K);
    return HIPT_E_UNSUPPORTED;
}
