// CLAM_SB / ABMIL gated-attention pooling, bf16 hot configuration [S0, 128, 64], streaming form
// (models/model_clam.py:41-64, 83-92, 147-183; same math as abmil.hip, which remains the general kernel).
//
// HBM-bound design: the 100 000 x 384 bf16 bag (76.8 MB) is read exactly once; nothing else moves.
//   * weights are staged ONCE per workgroup into LDS as swizzled MFMA operand images (W1: 6 slabs of
//     [128 hidden x 64 k] = 96 KiB; [Wa;Wb] with rows interleaved a,a,b,b: 32 KiB);
//   * every WAVE owns a contiguous range of rows end to end — no barrier, no LDS exchange in steady
//     state.  Per step of 32 rows (2 MFMA row fragments):
//       - the rows go HBM -> VGPRs directly in operand-fragment layout (lane (li,g) loads the 16-byte
//         chunks g, g+4, ... of row li), double buffered one step ahead;
//       - h1 = ReLU(x W1^T + b1) accumulates in registers (W1 fragments read from LDS, each feeding 2 MFMAs);
//       - the accumulators are re-packed in place as the operand of the gate GEMM (accumulator-as-operand:
//         the lane that owns 4 consecutive hidden units of a row owns exactly those K slots; the [Wa;Wb]
//         fragments are read with the matching K permutation), so h1 never leaves the register file;
//       - tanh * sigmoid as two sigmoids, dotted with wc: lane-local sum over the gate units + 2 shuffles;
//       - softmax pooling in fp32 on the un-rounded h1 accumulators: per-wave running max, each lane
//         accumulates its own rows' p * h1, one 16-lane reduction at the very end.
//   * partial (max, sum, acc[128]) per wave; the shared combine kernel merges them.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int SLAB = 128 * 128;  // 16 KiB: 128 rows x 64 bf16
constexpr int S1 = 128, S2 = 64;
constexpr float LOG2E = 1.4426950408889634f;

// tanh(x) * sigmoid(y) * w with ONE reciprocal: (E - 1) w / ((E + 1)(1 + F)), E = e^{2x}, F = e^{-y}.
// x is clamped to +-15 (tanh is 1 - 2e-13 there) so E stays finite; F = inf gives 0 as it should.
__device__ __forceinline__ float gate1(float x, float y, float w) {
    x = fminf(fmaxf(x, -15.0f), 15.0f);
    const float E = __builtin_amdgcn_exp2f(x * (2.0f * LOG2E));
    const float F = __builtin_amdgcn_exp2f(y * -LOG2E);
    return (E - 1.0f) * w * __builtin_amdgcn_rcpf((E + 1.0f) * (1.0f + F));
}

__device__ __forceinline__ float sigm(float x) {  // 1 / (1 + 2^(-x log2 e)); saturates cleanly at +-inf
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -LOG2E));
}

// 16 bytes at an 8-byte-aligned offset through a buffer resource, sc1 (bypasses the L1: see the fused combine)
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, int off) {
    const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16), hi = __builtin_amdgcn_raw_buffer_load_b64(r, off + 8, 0, 16);
    // (bit-cast the whole vectors: __builtin_bit_cast on a vector-element lvalue reads element 0, common.h)
    const f32x2 l = __builtin_bit_cast(f32x2, lo), h = __builtin_bit_cast(f32x2, hi);
    return f32x4{l[0], l[1], h[0], h[1]};
}

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define DSR64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define LGKM(n)                                             \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)

// End of a workgroup's stream: 16-lane reduction of the per-lane pools, merge of the 4 waves through LDS, partial
// (max, sum, acc[128]) to HBM and -- with a ticket -- the fused combine by the workgroup that finishes last.
__device__ __forceinline__ void finish_bag(char* smem, f32x4 (&pool)[8], float m_run, float l_lane, int nstep,
                                           float* __restrict__ partials, unsigned* __restrict__ ticket,
                                           const float* __restrict__ wcls, const float* __restrict__ bcls, int C,
                                           float* __restrict__ M, float* __restrict__ logits, float* __restrict__ Y_prob,
                                           int64_t* __restrict__ Y_hat) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    // every lane (li, g) holds the contribution of its rows to columns 16nf + 4g + e: sum over li (16 lanes);
    // l: the 4 g-lanes of a row hold the same p, count each row once (g == 0) and sum over li
    float l = g == 0 ? l_lane : 0.f;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        l += __shfl_xor(l, o, 64);
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[nf][e] += __shfl_xor(pool[nf][e], o, 64);
    }
    // merge the 4 waves of this workgroup through LDS (the weight images are dead now): slot w = (m, l, acc[128])
    __syncthreads();
    float* red = (float*)smem;
    if (lane == 0) {
        red[wave * 132] = nstep > 0 ? m_run : -INFINITY;
        red[wave * 132 + 1] = l;
    }
    if (li == 0) {
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave * 132 + 2 + 16 * nf + 4 * g + e] = pool[nf][e];
    }
    __syncthreads();
    if (tid < S1 + 2) {
        const float m0 = red[0], m1 = red[132], m2 = red[264], m3 = red[396];
        const float mm = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        float* pw = partials + (int64_t)blockIdx.x * (2 + S1);
        // (agent-scope relaxed stores = global_store sc0 sc1: they leave the XCD's L2, so the merging workgroup can
        //  read them with sc1 loads and no fence -- a device-scope release would write back the whole L2, 12 us here)
        if (tid == 0) {
            __hip_atomic_store(&pw[0], mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const float f0 = mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m0 - mm) * LOG2E);
            const float f1 = mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m1 - mm) * LOG2E);
            const float f2 = mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m2 - mm) * LOG2E);
            const float f3 = mm == -INFINITY ? 0.f : __builtin_amdgcn_exp2f((m3 - mm) * LOG2E);
            __hip_atomic_store(&pw[tid], red[tid] * f0 + red[132 + tid] * f1 + red[264 + tid] * f2 + red[396 + tid] * f3, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- fused combine (model_clam.py:180-183): the workgroup whose ticket is the last one merges all partials,
    //      applies the bag classifier, softmax and argmax -- no second launch.  Hand-off without fences
    //      (MI355X_MICROARCH.md, hand-off table row 1): sc1 stores, every storing wave waits vmcnt(0), workgroup
    //      barrier, ONE agent-scope atomic per workgroup; the workgroup whose
    //      add came last reads with sc1 loads after a workgroup barrier.
    if (ticket) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)(red + 600);
        if (tid == 0) {
            const bool last = atomicAdd(ticket, 1u) == gridDim.x - 1;
            *flag = last;
            // the ticket is an arrival COUNTER that starts at zero (hipt_clam_sb_forward's contract: the caller zeroes the
            // ticket block once, when it allocates the workspace) and that the last arriver puts back to zero: correct for
            // any dispatch order and any mix of streams, replayable from a graph, no memset node per call
            if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (*flag) {
            const int G = gridDim.x, stride = 2 + S1;
            float* Fs = red + 640;    // [256] rescale factors
            float* Cs = red + 1024;   // [8][128] column partial sums
            float* Ms = red + 2048;   // [128]
            float* Ls = red + 2176;   // [C <= 64]
            float* wr = red + 2240;   // [4] wave reductions
            // every load of the partials is an sc1 load (buffer loads with the sc1 cache-policy bit)
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)partials, 0, G * stride * 4, 0x00020000);
            constexpr int SC1 = 16;
            // all of this thread's share of the partials is requested at once, before anything is reduced: ONE memory
            // round trip (in batches of 8 behind the max / factor computation it was five).  32 threads x 16 B cover
            // one partial row, 8 rows per pass; rows past G are out of the buffer's range and read as zero
            const int c4 = tid & 31, part = tid >> 5;
            f32x4 rowv[32];  // (rows are 520 B apart: 8-byte aligned -> two 8-byte loads each)
#pragma unroll
            for (int i = 0; i < 32; ++i) rowv[i] = ld4(prs, ((part + 8 * i) * stride + 2 + 4 * c4) * 4);
            f32x2 ml = {-INFINITY, 0.f};  // (max, sum)
            if (tid < G) {
                ml[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, tid * stride * 4, 0, SC1));
                ml[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, tid * stride * 4 + 4, 0, SC1));
            }
            float mx = wave_max(ml[0]);
            if (lane == 0) wr[wave] = mx;
            __syncthreads();
            mx = fmaxf(fmaxf(wr[0], wr[1]), fmaxf(wr[2], wr[3]));
            __syncthreads();
            const float f = tid < G ? expf(ml[0] - mx) : 0.f;
            Fs[tid] = f;
            float ls = wave_sum(ml[1] * f);
            if (lane == 0) wr[wave] = ls;
            __syncthreads();
            const float L = wr[0] + wr[1] + wr[2] + wr[3];
            {   // column sums, in row order (Fs is 0 past G)
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 32; ++i) a += f32x4{0.f, 0.f, 0.f, 0.f} + rowv[i] * Fs[part + 8 * i];
                *(f32x4*)(Cs + part * S1 + 4 * c4) = a;
            }
            __syncthreads();
            if (tid < S1) {
                float a = 0.f;
#pragma unroll
                for (int part = 0; part < 8; ++part) a += Cs[part * S1 + tid];
                a /= L;
                Ms[tid] = a;
                M[tid] = a;
            }
            __syncthreads();
            for (int k = wave; k < C; k += 4) {
                float a = Ms[lane] * wcls[(int64_t)k * S1 + lane] + Ms[lane + 64] * wcls[(int64_t)k * S1 + lane + 64];
                a = wave_sum(a);
                if (lane == 0) Ls[k] = a + bcls[k];
            }
            __syncthreads();
            if (tid == 0) {
                float lm = -INFINITY;
                int arg = 0;
                for (int k = 0; k < C; ++k)
                    if (Ls[k] > lm) {
                        lm = Ls[k];
                        arg = k;
                    }
                float se = 0.f;
                for (int k = 0; k < C; ++k) se += expf(Ls[k] - lm);
                for (int k = 0; k < C; ++k) {
                    logits[k] = Ls[k];
                    Y_prob[k] = expf(Ls[k] - lm) / se;
                }
                Y_hat[0] = arg;
            }
        }
    }
}

template <int KS>  // S0 = 64 * KS
__global__ __launch_bounds__(256, 1) void abmil_stream_kernel(const bf16_t* __restrict__ bag, int N, int rows_per_wave,
                                                              const bf16_t* __restrict__ w1, const float* __restrict__ b1,
                                                              const bf16_t* __restrict__ wab, const float* __restrict__ bab,
                                                              const float* __restrict__ wc, const float* __restrict__ bc,
                                                              float* __restrict__ A_raw, float* __restrict__ partials,
                                                              int attention_only, unsigned long long* stamps,
                                                              unsigned* __restrict__ ticket, const float* __restrict__ wcls,
                                                              const float* __restrict__ bcls, int C, float* __restrict__ M,
                                                              float* __restrict__ logits, float* __restrict__ Y_prob,
                                                              int64_t* __restrict__ Y_hat) {
#define ASTAMP(k)                                                                                          \
    do {                                                                                                   \
        if (HIPT_STAMPS_ON(stamps) && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    ASTAMP(0);
    // (finish ticket of the fused combine: zero on entry -- the caller zeroes it once, the last arriver of every launch
    //  puts it back to zero, see finish_bag -- so nothing depends on which workgroup is dispatched first)
    constexpr int S0 = KS * 64;
    constexpr int NC = KS * 2;  // 16-byte chunks per lane per row
    extern __shared__ __attribute__((aligned(16))) char smem[];  // W1 image (KS slabs) | [Wa;Wb] image (2 slabs)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;

    // ---- my rows ----
    const int gw = blockIdx.x * 4 + wave;
    const int rbeg = gw * rows_per_wave;
    int rend = rbeg + rows_per_wave;
    rend = rend < N ? rend : N;
    const int nstep = rend > rbeg ? (rend - rbeg + 31) / 32 : 0;

    auto load_rows = [&](u32x4 (&xf)[2][NC], int s) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            int r = rbeg + s * 32 + m * 16 + li;
            r = r < rend ? r : rend - 1;
            const bf16_t* xr = bag + (int64_t)r * S0;
#pragma unroll
            for (int c = 0; c < NC; ++c) xf[m][c] = *(const u32x4*)(xr + (g + 4 * c) * 8);
        }
    };

    // the first rows are requested BEFORE the weights: both latencies overlap (the loads were 5.7 us of a 36 us kernel)
    u32x4 xa[2][NC], xb[2][NC];
    if (nstep > 0) load_rows(xa, 0);

    // ---- stage the weights (LDS-DMA, swizzle on the source address) ----
    {
        const int r0 = wave * 8 + (lane >> 3);
        const int ch0 = (lane & 7) ^ ((r0 >> 1) & 7);
#pragma unroll
        for (int kt = 0; kt < KS; ++kt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                glds16(w1 + (int64_t)(q * 32 + r0) * S0 + (kt * 8 + ch0) * 8, smem + kt * SLAB + (q * 4 + wave) * 1024);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = q * 32 + r0;  // packed gate row -> source row ((r&3)>>1)*S2 + (r>>2)*2 + (r&1)
                const int srow = ((r & 3) >> 1) * S2 + (r >> 2) * 2 + (r & 1);
                glds16(wab + (int64_t)srow * S1 + (sl * 8 + ch0) * 8, smem + (KS + sl) * SLAB + (q * 4 + wave) * 1024);
            }
    }
    // small constants -> LDS (kept out of the register file: the streaming loop needs all of it):
    // b1[128] | ba[64] | bb[64] | wc[64]
    float* cst = (float*)(smem + (KS + 2) * SLAB);
    for (int i = tid; i < S1; i += 256) cst[i] = b1[i];
    for (int i = tid; i < 2 * S2; i += 256) cst[S1 + i] = bab[i];
    for (int i = tid; i < S2; i += 256) cst[S1 + 2 * S2 + i] = wc[i];
    const float bcv = bc[0];
    wait_vm0();
    __syncthreads();  // weights are in LDS; from here on the waves never synchronise again
    ASTAMP(1);

    const uint32_t lbase = lds_addr(smem);
    uint32_t foff[2], fhi[2], f2off[2][2];  // ds_read immediates are 16-bit: slabs 3.. use a second base
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        foff[ks] = lbase + li * 128 + (((g + 4 * ks) ^ ((lane >> 1) & 7)) << 4);
        fhi[ks] = foff[ks] + 3 * SLAB;
    }
#pragma unroll
    for (int fl = 0; fl < 2; ++fl)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int byte = 64 * fl + 32 * h + 8 * g;  // K slots 32fl + 4g + (0..3) and + 16 of a 64-wide slab
            f2off[fl][h] = lbase + KS * SLAB + li * 128 + (((byte >> 4) ^ ((lane >> 1) & 7)) << 4) + (byte & 8);
        }

    float m_run = -INFINITY, l_lane = 0.f;
    f32x4 pool[8];
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) pool[nf] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](u32x4 (&xf)[2][NC], int s) {
        // ================= phase 1: acc1[m][nf] = x W1^T  (lane: row li of fragment m, hidden 16nf + 4g + e) =================
        f32x4 acc1[2][8];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) acc1[m][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#define RD4(w, addr, base)            \
    DSR128(w[0], addr, base + 0);     \
    DSR128(w[1], addr, base + 2048);  \
    DSR128(w[2], addr, base + 4096);  \
    DSR128(w[3], addr, base + 6144)
#define MM1(w, c, q0)                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                     \
        Tr<bf16_t>::mma16(acc1[0][(q0) + j], w[j], xf[0][c]);           \
        Tr<bf16_t>::mma16(acc1[1][(q0) + j], w[j], xf[1][c]);           \
    }                                                                   \
    __builtin_amdgcn_sched_barrier(0)
// one W1 slab (kt): wa holds (kt, ks0, nf 0-3) on entry; B = base pair for this slab, O = its byte offset (< 64 KiB)
#define PH1(B, O, c0)                                   \
    RD4(wb, B[0], O + 8192);                             \
    LGKM(4); MM1(wa, c0, 0);                             \
    RD4(wa, B[1], O);                                    \
    LGKM(4); MM1(wb, c0, 4);                             \
    RD4(wb, B[1], O + 8192);                             \
    LGKM(4); MM1(wa, (c0) + 1, 0)
        {
            u32x4 wa[4], wb[4];
            RD4(wa, foff[0], 0);
            PH1(foff, 0, 0);     RD4(wa, foff[0], 16384); LGKM(4); MM1(wb, 1, 4);
            PH1(foff, 16384, 2); RD4(wa, foff[0], 32768); LGKM(4); MM1(wb, 3, 4);
            if constexpr (KS == 3) {
                PH1(foff, 32768, 4); LGKM(0); MM1(wb, 5, 4);
            } else {
                PH1(foff, 32768, 4);  RD4(wa, fhi[0], 0);     LGKM(4); MM1(wb, 5, 4);
                PH1(fhi, 0, 6);       RD4(wa, fhi[0], 16384); LGKM(4); MM1(wb, 7, 4);
                PH1(fhi, 16384, 8);   RD4(wa, fhi[0], 32768); LGKM(4); MM1(wb, 9, 4);
                PH1(fhi, 32768, 10);  LGKM(0); MM1(wb, 11, 4);
            }
        }
#undef PH1
#undef RD4
#undef MM1
        // ================= h1 = ReLU(. + b1); re-pack as operand fragments of the gate GEMM =================
        u32x4 hf[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int nf = 0; nf < 8; ++nf) {
                acc1[m][nf] += *(const f32x4*)(cst + 16 * nf + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[m][nf][e] = fmaxf(acc1[m][nf][e], 0.f);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                u32x4 o;
                o[0] = pack_bf16x2(acc1[m][2 * f][0], acc1[m][2 * f][1]);
                o[1] = pack_bf16x2(acc1[m][2 * f][2], acc1[m][2 * f][3]);
                o[2] = pack_bf16x2(acc1[m][2 * f + 1][0], acc1[m][2 * f + 1][1]);
                o[3] = pack_bf16x2(acc1[m][2 * f + 1][2], acc1[m][2 * f + 1][3]);
                hf[m][f] = o;
            }
        }
        // ================= phase 2: ab[m][n2] = h1 [Wa;Wb]^T, gate, A =================
        f32x4 acc2[2][8];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) acc2[m][n2] = f32x4{0.f, 0.f, 0.f, 0.f};
#define RD8(l, h, alo, ahi, base)                                   \
    DSR64(l[0], alo, base + 0);    DSR64(h[0], ahi, base + 0);      \
    DSR64(l[1], alo, base + 2048); DSR64(h[1], ahi, base + 2048);   \
    DSR64(l[2], alo, base + 4096); DSR64(h[2], ahi, base + 4096);   \
    DSR64(l[3], alo, base + 6144); DSR64(h[3], ahi, base + 6144)
#define MM2(l, h, f, q0)                                                        \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                             \
        u32x4 wf;                                                               \
        wf[0] = l[j][0]; wf[1] = l[j][1]; wf[2] = h[j][0]; wf[3] = h[j][1];     \
        Tr<bf16_t>::mma16(acc2[0][(q0) + j], wf, hf[0][f]);                     \
        Tr<bf16_t>::mma16(acc2[1][(q0) + j], wf, hf[1][f]);                     \
    }                                                                           \
    __builtin_amdgcn_sched_barrier(0)
        {
            u32x2 la[4], ha[4], lb[4], hb[4];
            // operand fragment f = 2 * slab + fl covers hidden units [32f, 32f + 32)
            RD8(la, ha, f2off[0][0], f2off[0][1], 0);
            RD8(lb, hb, f2off[0][0], f2off[0][1], 8192);
            LGKM(8); MM2(la, ha, 0, 0);
            RD8(la, ha, f2off[1][0], f2off[1][1], 0);
            LGKM(8); MM2(lb, hb, 0, 4);
            RD8(lb, hb, f2off[1][0], f2off[1][1], 8192);
            LGKM(8); MM2(la, ha, 1, 0);
            RD8(la, ha, f2off[0][0], f2off[0][1], 16384);
            LGKM(8); MM2(lb, hb, 1, 4);
            RD8(lb, hb, f2off[0][0], f2off[0][1], 24576);
            LGKM(8); MM2(la, ha, 2, 0);
            RD8(la, ha, f2off[1][0], f2off[1][1], 16384);
            LGKM(8); MM2(lb, hb, 2, 4);
            RD8(lb, hb, f2off[1][0], f2off[1][1], 24576);
            LGKM(8); MM2(la, ha, 3, 0);
            LGKM(0); MM2(lb, hb, 3, 4);
        }
#undef RD8
#undef MM2
        float a_row[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float gs = 0.f;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const f32x4 v = acc2[m][n2];  // (a_j0, a_j0+1, b_j0, b_j0+1), j0 = 8 n2 + 2 g
                const int j0 = 8 * n2 + 2 * g;
                const f32x2 ba = *(const f32x2*)(cst + S1 + j0), bb = *(const f32x2*)(cst + S1 + S2 + j0),
                            cw = *(const f32x2*)(cst + S1 + 2 * S2 + j0);
                gs += gate1(v[0] + ba[0], v[2] + bb[0], cw[0]);
                gs += gate1(v[1] + ba[1], v[3] + bb[1], cw[1]);
            }
            gs += __shfl_xor(gs, 16, 64);
            gs += __shfl_xor(gs, 32, 64);
            const int r = rbeg + s * 32 + m * 16 + li;
            const bool valid = r < rend;
            a_row[m] = valid ? gs + bcv : -INFINITY;
            if (valid && g == 0) A_raw[r] = a_row[m];
        }
        if (attention_only) return;
        // ================= softmax pooling (fp32, on the un-rounded h1) =================
        const float mt = wave_max(fmaxf(a_row[0], a_row[1]));  // finite: every step has at least one valid row
        const float m_new = fmaxf(m_run, mt);
        const float resc = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);  // 0 on the first step
        m_run = m_new;
        const float p0 = __builtin_amdgcn_exp2f((a_row[0] - m_new) * LOG2E), p1 = __builtin_amdgcn_exp2f((a_row[1] - m_new) * LOG2E);
        l_lane = l_lane * resc + p0 + p1;
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) pool[nf] = pool[nf] * resc + acc1[0][nf] * p0 + acc1[1][nf] * p1;
    };

    // ---- double-buffered stream over my rows ----
    for (int s = 0; s < nstep; s += 2) {
        if (s + 1 < nstep) load_rows(xb, s + 1);
        if (s == 0) ASTAMP(2);
        compute(xa, s);
        if (s == 0) ASTAMP(3);
        if (s + 1 < nstep) {
            if (s + 2 < nstep) load_rows(xa, s + 2);
            compute(xb, s + 1);
            if (s == 0) ASTAMP(4);
        }
    }
    ASTAMP(5);

    if (!attention_only) finish_bag(smem, pool, m_run, l_lane, nstep, partials, ticket, wcls, bcls, C, M, logits, Y_prob, Y_hat);
    ASTAMP(6);
}


// ======================================================================================================================
// Software-pipelined form for S0 = 384 (the bench bag): the gate + pooling arithmetic of step s-1 (transcendental VALU
// work, as long as the MFMAs of phase 1) runs UNDER the phase-1 MFMAs of step s.
//   * phase 1 = 24 groups of 8 MFMAs (one W1 fragment set each, read one group ahead); group G < 16 carries gate unit G
//     (2 rows x 2 gate pairs) of the previous step, group 16 the row logits / running max, groups 16-23 the pooling
//     update of one 16-column block each;
//   * the h1 accumulators therefore live for two steps: two sets, alternated by unrolling the loop twice;
//   * the rows stream through ONE register image: the 16-byte chunks of k-step c are re-loaded for the next step as
//     soon as step s has used them (buffer loads: rows past the wave's range read as zero without HBM traffic), so
//     HBM requests spread over the whole step instead of bursting;
//   * b1 and [ba;bb] enter as the C operands of the first MFMAs, wc through the same counted-wait LDS reads as W1.
#define DSRN128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define DSRN64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define LGKMN(n)                                                \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
    __builtin_amdgcn_sched_barrier(0)

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// one gate unit: 2 gate pairs of one row; v = (a_j, a_j+1, b_j, b_j+1) with the biases already in, cw = wc[j], wc[j+1]
// (contraction off, here and where the results are summed: hipcc fuses `num * r` into the caller's sum in one inlined
//  instance and not in another, and a row's logit must not depend on the fragment / step it lands in)
__device__ __forceinline__ f32x2 gate_pair(const f32x4& v, const f32x2& cw) {
#pragma clang fp contract(off)
    f32x2 x = {__builtin_amdgcn_fmed3f(v[0], -15.0f, 15.0f), __builtin_amdgcn_fmed3f(v[1], -15.0f, 15.0f)};
    f32x2 y = {v[2], v[3]};
    x *= 2.0f * LOG2E;
    y *= -LOG2E;
    const f32x2 E = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
    const f32x2 F = {__builtin_amdgcn_exp2f(y[0]), __builtin_amdgcn_exp2f(y[1])};
    const f32x2 num = (E - 1.0f) * cw, den = (E + 1.0f) * (F + 1.0f);
    const f32x2 r = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
    return num * r;
}

__global__ __launch_bounds__(256, 1) void abmil_pipe_kernel(const bf16_t* __restrict__ bag, int N, int rows_per_wave,
                                                            const bf16_t* __restrict__ w1, const float* __restrict__ b1,
                                                            const bf16_t* __restrict__ wab, const float* __restrict__ bab,
                                                            const float* __restrict__ wc, const float* __restrict__ bc,
                                                            float* __restrict__ A_raw, float* __restrict__ partials,
                                                            int attention_only, unsigned long long* stamps, unsigned* __restrict__ ticket,
                                                            const float* __restrict__ wcls, const float* __restrict__ bcls, int C,
                                                            float* __restrict__ M, float* __restrict__ logits,
                                                            float* __restrict__ Y_prob, int64_t* __restrict__ Y_hat) {
    ASTAMP(0);
    constexpr int KS = 6, S0 = 384, NC = 12;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // W1 image (6 slabs) | [Wa;Wb] image (2 slabs) | constants

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;

    const int gw = blockIdx.x * 4 + wave;
    const int rbeg = gw * rows_per_wave;
    int rend = rbeg + rows_per_wave;
    rend = rend < N ? rend : N;
    const int nrows = rend > rbeg ? rend - rbeg : 0;
    const int nstep = (nrows + 31) / 32;

    // my rows through a buffer resource that ends with them: a chunk of a row past the end reads as zero, no traffic
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(bag + (int64_t)(nrows ? rbeg : 0) * S0), 0, nrows * S0 * 2, 0x00020000);
    int voff[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) voff[m] = (m * 16 + li) * (S0 * 2) + g * 16;
    u32x4 xf[2][NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int m = 0; m < 2; ++m) xf[m][c] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[m] + c * 64, 0, 0);

    // ---- stage the weights (LDS-DMA, swizzle on the source address) ----
    {
        const int r0 = wave * 8 + (lane >> 3);
        const int ch0 = (lane & 7) ^ ((r0 >> 1) & 7);
#pragma unroll
        for (int kt = 0; kt < KS; ++kt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                glds16(w1 + (int64_t)(q * 32 + r0) * S0 + (kt * 8 + ch0) * 8, smem + kt * SLAB + (q * 4 + wave) * 1024);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = q * 32 + r0;  // packed gate row -> source row ((r&3)>>1)*S2 + (r>>2)*2 + (r&1)
                const int srow = ((r & 3) >> 1) * S2 + (r >> 2) * 2 + (r & 1);
                glds16(wab + (int64_t)srow * S1 + (sl * 8 + ch0) * 8, smem + (KS + sl) * SLAB + (q * 4 + wave) * 1024);
            }
    }
    // constants -> LDS: b1[128] | gate bias in accumulator order [n2][g][(ba_j, ba_j+1, bb_j, bb_j+1)], j = 8 n2 + 2 g | wc[64]
    float* cst = (float*)(smem + (KS + 2) * SLAB);
    if (tid < S1) {
        cst[tid] = b1[tid];
        const int j = 8 * (tid >> 4) + 2 * ((tid >> 2) & 3) + (tid & 1);
        cst[S1 + tid] = bab[(tid & 2) ? S2 + j : j];
    } else if (tid < S1 + S2) {
        cst[S1 + tid] = wc[tid - S1];
    }
    const float bcv = bc[0];
    wait_vm0();
    __syncthreads();  // weights are in LDS; from here on the waves never synchronise again
    ASTAMP(1);

    const uint32_t lbase = lds_addr(smem);
    uint32_t foff[2], fhi[2], f2off[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        foff[ks] = lbase + li * 128 + (((g + 4 * ks) ^ ((lane >> 1) & 7)) << 4);
        fhi[ks] = foff[ks] + 3 * SLAB;
    }
#pragma unroll
    for (int fl = 0; fl < 2; ++fl)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int byte = 64 * fl + 32 * h + 8 * g;
            f2off[fl][h] = lbase + KS * SLAB + li * 128 + (((byte >> 4) ^ ((lane >> 1) & 7)) << 4) + (byte & 8);
        }
    const uint32_t cwaddr = lbase + (KS + 2) * SLAB + (2 * S1 + 2 * g) * 4;  // wc[8 n2 + 2 g]: + 32 n2 bytes
    const float* b1f = cst + 4 * g;                                            // b1[16 nf + 4 g ..]: + 16 nf floats
    const float* gbf = cst + S1 + 4 * g;                                       // gate bias of (n2, g): + 16 n2 floats

    float m_run = -INFINITY, l_lane = 0.f;
    bool m1 = true;
    f32x4 pool[8];
#pragma unroll
    for (int nf = 0; nf < 8; ++nf) pool[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accA[2][8], accB[2][8], acc2[2][8];
    u32x4 hf[2][4];
    f32x2 gs2[2];   // per-row logit partial sums of the step being gated
    float p_row[2]; // softmax weights of its two rows
    float resc = 0.f;

    // logits of step sp from the finished gate sums; running max, rescale factor, the two softmax weights
    auto gate_finish = [&](int sp) __attribute__((always_inline)) {
        float a_row[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float gs = gs2[m][0] + gs2[m][1];
            gs += __shfl_xor(gs, 16, 64);
            gs += __shfl_xor(gs, 32, 64);
            const int r = sp * 32 + m * 16 + li;
            const bool valid = r < nrows;
            a_row[m] = valid ? gs + bcv : -INFINITY;
            if (valid && g == 0) A_raw[rbeg + r] = a_row[m];
        }
        const float mt = wave_max(fmaxf(a_row[0], a_row[1]));  // finite: every step has at least one valid row
        const float m_new = fmaxf(m_run, mt);
        resc = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);  // 0 on the first step
        m_run = m_new;
        p_row[0] = __builtin_amdgcn_exp2f((a_row[0] - m_new) * LOG2E);
        p_row[1] = __builtin_amdgcn_exp2f((a_row[1] - m_new) * LOG2E);
        l_lane = l_lane * resc + p_row[0] + p_row[1];
    };

    // phase 1 of step s into cur; with PIPE the gate / pooling of step s - 1 (acc2, prev) rides along
    auto phase1 = [&](auto PIPE_, f32x4 (&cur)[2][8], f32x4 (&prev)[2][8], int s) __attribute__((always_inline)) {
        constexpr bool PIPE = decltype(PIPE_)::value;
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) {
            const f32x4 b = *(const f32x4*)(b1f + 16 * nf);
            cur[0][nf] = b;
            cur[1][nf] = b;
        }
        if constexpr (PIPE) gs2[0] = gs2[1] = f32x2{0.f, 0.f};
        // (the step offset travels in the VGPR offset: that is the part the range check certainly covers)
        const int vnext[2] = {voff[0] + (s + 1) * 32 * S0 * 2, voff[1] + (s + 1) * 32 * S0 * 2};
        u32x4 w[2][4];
        f32x2 cw[2];
        const uint32_t a00 = foff[0], a01 = foff[1], a10 = fhi[0], a11 = fhi[1], cwa = cwaddr;
        // fragment set of group G: slab kt = G >> 2, k-step half ks = (G >> 1) & 1, hidden half q = G & 1
#define RDG(G)                                                                       \
    do {                                                                             \
        constexpr int kt_ = (G) >> 2, ks_ = ((G) >> 1) & 1, q_ = (G)&1;              \
        constexpr int o_ = (kt_ % 3) * 16384 + q_ * 8192;                            \
        const uint32_t ad_ = kt_ < 3 ? (ks_ ? a01 : a00) : (ks_ ? a11 : a10);        \
        DSRN128(w[(G)&1][0], ad_, o_);                                               \
        DSRN128(w[(G)&1][1], ad_, o_ + 2048);                                        \
        DSRN128(w[(G)&1][2], ad_, o_ + 4096);                                        \
        DSRN128(w[(G)&1][3], ad_, o_ + 6144);                                        \
        if constexpr (PIPE && (G) < 16) DSRN64(cw[(G)&1], cwa, ((G)&7) * 32);        \
    } while (0)
        RDG(0);
        sfor<0, 24>([&](auto G_) __attribute__((always_inline)) {
            constexpr int G = decltype(G_)::value;
            constexpr int c = G >> 1, q = G & 1;
            // fragment set G + 1 is requested before set G is waited for (counted: only its reads may be outstanding)
            if constexpr (G + 1 < 24) {
                RDG(G + 1);
                if constexpr (PIPE && G + 1 < 16) {
                    LGKMN(5);
                } else {
                    LGKMN(4);
                }
            } else {
                LGKMN(0);
            }
            if constexpr (PIPE) {
#pragma clang fp contract(off)
                if constexpr (G < 16) {
                    gs2[G >> 3] += gate_pair(acc2[G >> 3][G & 7], cw[G & 1]);
                } else {
                    if constexpr (G == 16) gate_finish(s - 1);
                    constexpr int nf = G - 16;
                    pool[nf] = pool[nf] * resc + prev[0][nf] * p_row[0] + prev[1][nf] * p_row[1];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) Tr<bf16_t>::mma16(cur[0][4 * q + j], w[G & 1][j], xf[0][c]);
            if (m1) {  // (the second fragment of a wave's last step may be empty: half the MFMAs of that step)
#pragma unroll
                for (int j = 0; j < 4; ++j) Tr<bf16_t>::mma16(cur[1][4 * q + j], w[G & 1][j], xf[1][c]);
            }
            if constexpr (q == 1) {  // k-step c is done with its rows: request the next step's
                xf[0][c] = __builtin_amdgcn_raw_buffer_load_b128(rs, vnext[0] + c * 64, 0, 0);
                xf[1][c] = __builtin_amdgcn_raw_buffer_load_b128(rs, vnext[1] + c * 64, 0, 0);
            }
            if constexpr (PIPE) {  // an MFMA leaves 8 of its 16 cycles to other vector instructions: deal them out evenly
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#undef RDG
    };

    // h1 = ReLU(cur) in place (fp32, for the pooling one step later) and as operand fragments of the gate GEMM
    auto relu_pack = [&](f32x4 (&cur)[2][8]) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int e = 0; e < 4; ++e) cur[m][nf][e] = fmaxf(cur[m][nf][e], 0.f);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                u32x4 o;
                o[0] = pack_bf16x2(cur[m][2 * f][0], cur[m][2 * f][1]);
                o[1] = pack_bf16x2(cur[m][2 * f][2], cur[m][2 * f][3]);
                o[2] = pack_bf16x2(cur[m][2 * f + 1][0], cur[m][2 * f + 1][1]);
                o[3] = pack_bf16x2(cur[m][2 * f + 1][2], cur[m][2 * f + 1][3]);
                hf[m][f] = o;
            }
        }
    };

    // phase 2: acc2 = [ba;bb] + h1 [Wa;Wb]^T (8 groups of 8 MFMAs; fragment f = 2 slab + fl covers hidden [32f, 32f+32))
    auto phase2 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const f32x4 b = *(const f32x4*)(gbf + 16 * n2);
            acc2[0][n2] = b;
            acc2[1][n2] = b;
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x2 wl[2][4], wh[2][4];
        const uint32_t b00 = f2off[0][0], b01 = f2off[0][1], b10 = f2off[1][0], b11 = f2off[1][1];
#define RD2(G)                                                                     \
    do {                                                                           \
        constexpr int f_ = (G) >> 1, q_ = (G)&1;                                   \
        constexpr int o_ = (f_ >> 1) * 16384 + q_ * 8192;                          \
        const uint32_t lo_ = (f_ & 1) ? b10 : b00, hi_ = (f_ & 1) ? b11 : b01;     \
        DSRN64(wl[(G)&1][0], lo_, o_);        DSRN64(wh[(G)&1][0], hi_, o_);        \
        DSRN64(wl[(G)&1][1], lo_, o_ + 2048); DSRN64(wh[(G)&1][1], hi_, o_ + 2048); \
        DSRN64(wl[(G)&1][2], lo_, o_ + 4096); DSRN64(wh[(G)&1][2], hi_, o_ + 4096); \
        DSRN64(wl[(G)&1][3], lo_, o_ + 6144); DSRN64(wh[(G)&1][3], hi_, o_ + 6144); \
    } while (0)
        RD2(0);
        sfor<0, 8>([&](auto G_) __attribute__((always_inline)) {
            constexpr int G = decltype(G_)::value;
            constexpr int f = G >> 1, q = G & 1;
            if constexpr (G + 1 < 8) {
                RD2(G + 1);
                LGKMN(8);
            } else {
                LGKMN(0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 wf;
                wf[0] = wl[G & 1][j][0]; wf[1] = wl[G & 1][j][1]; wf[2] = wh[G & 1][j][0]; wf[3] = wh[G & 1][j][1];
                Tr<bf16_t>::mma16(acc2[0][4 * q + j], wf, hf[0][f]);
                if (m1) Tr<bf16_t>::mma16(acc2[1][4 * q + j], wf, hf[1][f]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#undef RD2
    };

    // gate + pooling of the last step: nothing left to hide it under
    auto drain = [&](f32x4 (&prev)[2][8], int sp) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        // (all 16 units first, then the sums: summed as it goes, the chain of transcendentals runs at its latency)
        f32x2 u[2][8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) {
            const f32x2 cwv = *(const f32x2*)(cst + 2 * S1 + 8 * n2 + 2 * g);
            u[0][n2] = gate_pair(acc2[0][n2], cwv);
            u[1][n2] = m1 ? gate_pair(acc2[1][n2], cwv) : f32x2{0.f, 0.f};  // (an empty fragment: its rows are masked below)
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {  // same order of additions as the overlapped form: bit-identical logits
            gs2[m] = f32x2{0.f, 0.f};
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) gs2[m] += u[m][n2];
        }
        gate_finish(sp);
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) pool[nf] = pool[nf] * resc + prev[0][nf] * p_row[0] + prev[1][nf] * p_row[1];
    };

    constexpr std::false_type PLAIN{};
    constexpr std::true_type OVERLAP{};
    // m1: the step about to run has rows in its second fragment (false only for a last step of <= 16 rows)
    auto set_m1 = [&](int s) { m1 = s * 32 + 16 < nrows; };
    if (nstep > 0) {
        ASTAMP(2);
        set_m1(0);
        phase1(PLAIN, accA, accB, 0);
        ASTAMP(3);
        relu_pack(accA);
        phase2();
        ASTAMP(4);
        int s = 1;
        for (; s + 1 < nstep; s += 2) {
            phase1(OVERLAP, accB, accA, s);  // (m1 stays true: this is not the wave's last step)
            relu_pack(accB);
            phase2();
            set_m1(s + 1);
            phase1(OVERLAP, accA, accB, s + 1);
            relu_pack(accA);
            phase2();
        }
        if (s < nstep) {
            set_m1(s);
            phase1(OVERLAP, accB, accA, s);
            relu_pack(accB);
            phase2();
            drain(accB, s);
        } else {
            drain(accA, s - 1);
        }
    }
    ASTAMP(5);
    // (attention_only runs the same arithmetic -- its logits are bit-identical to a full forward's -- and stops here)
    if (!attention_only) finish_bag(smem, pool, m_run, l_lane, nstep, partials, ticket, wcls, bcls, C, M, logits, Y_prob, Y_hat);
    ASTAMP(6);
}

template <int KS>
int launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* partials,
           int* n_partials, unsigned* ticket, float* M, float* logits, float* Y_prob, int64_t* Y_hat, hipStream_t st) {
    constexpr int lds = (KS + 2) * SLAB + (S1 + 3 * S2) * 4;
    constexpr bool piped = KS == 6;  // S0 = 384: the software-pipelined form; S0 = 192: the plain streaming form
    HIPT_CUR_DEVICE(dev);
    // contiguous row ranges per wave; at most 256 workgroups x 4 waves
    int rows = (N + 1023) / 1024;
    // (rows per wave rounded to 2, not to a 16-row fragment: 100 000 rows are then 98 per wave on all 256 CUs instead of 112 on 224 --
    //  the same time, measured both ways; rows are masked one by one, a buffer resource ends each wave's range)
    rows = (rows + 1) / 2 * 2;
    const int waves = (N + rows - 1) / rows;
    const int grid = (waves + 3) / 4;
#ifdef HIPT_DEBUG_STAMPS  // diagnostic builds only (make DEBUG_STAMPS=1): the release library never allocates or synchronises
    static const bool want_stamps = getenv("HIPT_ABMIL_STAMPS") != nullptr;
    static unsigned long long* dbuf = nullptr;
    if (want_stamps && !dbuf) (void)hipMalloc(&dbuf, 512 * 8 * sizeof(unsigned long long));
#else
    constexpr bool want_stamps = false;
    constexpr unsigned long long* dbuf = nullptr;
#endif
    const bool fuse = !attention_only && ticket && M && w->n_classes <= 64 && grid <= 256;
    if constexpr (piped) {
        auto kp = abmil_pipe_kernel;
        static DevOnce once_p;
        if (!once_p.done[dev]) {
            if (hipFuncSetAttribute((const void*)kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
                hipt_set_error("hipFuncSetAttribute(abmil pipe) failed");
                return HIPT_E_LAUNCH;
            }
            once_p.done[dev] = true;
        }
        hipLaunchKernelGGL(kp, dim3(grid), dim3(256), lds, st, (const bf16_t*)bag, N, rows, (const bf16_t*)w->w1, w->b1,
                           (const bf16_t*)w->wab, w->bab, w->wc, w->bc, A_raw, partials, attention_only,
                           want_stamps ? dbuf : nullptr, fuse ? ticket : nullptr, w->wcls, w->bcls, w->n_classes, M, logits, Y_prob, Y_hat);
    } else {
        auto k = abmil_stream_kernel<KS>;
        static DevOnce once;
        if (!once.done[dev]) {
            if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
                hipt_set_error("hipFuncSetAttribute(abmil stream) failed");
                return HIPT_E_LAUNCH;
            }
            once.done[dev] = true;
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, (const bf16_t*)bag, N, rows, (const bf16_t*)w->w1, w->b1,
                           (const bf16_t*)w->wab, w->bab, w->wc, w->bc, A_raw, partials, attention_only, want_stamps ? dbuf : nullptr,
                           fuse ? ticket : nullptr, w->wcls, w->bcls, w->n_classes, M, logits, Y_prob, Y_hat);
    }
    HIPT_CHECK_LAUNCH();
#ifdef HIPT_DEBUG_STAMPS
    if (want_stamps && grid <= 512) {
        static unsigned long long h[512 * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbuf, (size_t)grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t6 = 0;
        for (int b = 0; b < grid; ++b) {
            if (h[b * 8] < t0) t0 = h[b * 8];
            if (h[b * 8 + 6] > t6) t6 = h[b * 8 + 6];
        }
        double ph[6] = {0, 0, 0, 0, 0, 0}, smax = 0;
        for (int b = 0; b < grid; ++b) {
            for (int k2 = 0; k2 < 6; ++k2) ph[k2] += (double)(h[b * 8 + k2 + 1] - h[b * 8 + k2]) * 0.01 / grid;
            const double s0 = (double)(h[b * 8] - t0) * 0.01;
            if (s0 > smax) smax = s0;
        }
        fprintf(stderr, "[abmil %s N=%d grid=%d rows/wave=%d] total %.1f us | start<=%.1f; weights->LDS %.1f; %s %.1f / %.1f / %.1f / %.1f; final %.1f\n",
                piped ? "pipe" : "stream", N, grid, rows, (double)(t6 - t0) * 0.01, smax, ph[0],
                piped ? "- / step 0 phase 1 / its gate GEMM / steps 1.. + drain:" : "issue loads / step 0 / step 1 / steps 2..:", ph[1], ph[2], ph[3],
                ph[4], ph[5]);
    }
#endif
    *n_partials = fuse ? 0 : grid;  // 0: the kernel has already produced M / logits / Y_prob / Y_hat
    return HIPT_OK;
}

}  // namespace

bool hipt_clam_stream_supported(const hipt_clam_weights* w) {
    return w->dtype == HIPT_BF16 && w->s1 == S1 && w->s2 == S2 && (w->s0 == 384 || w->s0 == 192) && !hipt_generic_only();
}

int hipt_clam_stream_launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw,
                            float* partials, int* n_partials, unsigned* ticket, float* M, float* logits, float* Y_prob,
                            int64_t* Y_hat, hipStream_t st) {
    if (w->s0 == 384) return launch<6>(w, bag, N, attention_only, A_raw, partials, n_partials, ticket, M, logits, Y_prob, Y_hat, st);
    if (w->s0 == 192) return launch<3>(w, bag, N, attention_only, A_raw, partials, n_partials, ticket, M, logits, Y_prob, Y_hat, st);
    hipt_set_error("clam stream: unsupported S0=%d", w->s0);
    return HIPT_E_UNSUPPORTED;
}
