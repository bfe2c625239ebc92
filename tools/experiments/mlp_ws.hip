// FUSED MLP SUB-BLOCK for D = 384 (ViT-256), WAVE-SPECIALISED:   x <- x + y1 + fc2( GELU( fc1( LN2(x + y1) ) ) )
//   (Block.forward second half, HIPT_4K/vision_transformer.py:151 with Mlp.forward :98-104.)
//
// Why another form.  tools/issue_mix_probe.hip measured what one wave per SIMD pays for the things mlp_pipe.hip / mlp32.hip put
// between their MFMAs (cycles per group of four 32x32x16 MFMAs = 128 matrix-pipe cycles): 4 LDS fragment reads +14..34, one
// LDS-DMA piece +36..60, GELU arithmetic beyond the 24 issue cycles an MFMA leaves free at full cost -- the single-wave kernels
// run their chunk phases at ~230 cycles per group whatever the order.  With TWO waves per SIMD the same streams overlap
// (reads + DMA + 20 VALU: 152 instead of 235): one wave's issue stalls are the other wave's issue slots.
// The accumulators of 32 rows (fc1 tile 16 + fc2 192 registers) plus the activations (96) do not fit twice into a SIMD's 512
// registers, so the two waves of a SIMD share ONE set of 32 rows and split the work instead:
//   * F1 waves (0..3): LayerNorm-2 of x + y1 into the fc1 B operand (96 registers, kept in the accumulator file), then per STEP
//     k one 32-wide hidden tile: 24 MFMAs (acc 16 registers, double-buffered) and, in their gaps, the GELU of tile k-1 (un-packed
//     VALU, interleaved over the two elements of a unit) whose bf16 result IS the fc2 operand fragment: written to LDS (2 KiB
//     per wave and step).
//   * F2 waves (4..7, wave w+4 shares rows and SIMD with wave w): per step the fc2 MFMAs of hidden tile k-2 (2 k-steps x 12 output
//     tiles = 24 MFMAs into the 192-register accumulator, B operand = the two fragments F1 left in LDS), then the residual epilogue
//     and the next block's LayerNorm-1.
//   One s_barrier per step for all 8 waves.  F1's row phase of the NEXT tile runs beside F2's last step + epilogue of the current one.
//   * weights: one packed image, units of 24 fragments (24 KiB) in consumption order  W1(0) W1(1) | W1(2) W2(0) | ... | W1(47) W2(45)
//     | W2(46) | W2(47);  5 ring slots; every wave issues 3 of a unit's 24 LDS-DMA pieces, one piece per 4-MFMA group; the ring is
//     kept full (lookahead 2.5 steps), waits are counted.
// Registers: 256 per wave, ONE static split for both roles (hipcc: arch VGPRs | accumulator file): F1 keeps X and its tile
// accumulators in the accumulator file ("a" constraints), F2 keeps 8 of its 12 output tiles there and 4 in arch VGPRs (hipcc splits the 256 registers 128 | 128).
// Every MFMA / LDS access / VALU of the steps is volatile inline asm in program order (hipcc only allocates registers and forms
// addresses); hazards the assembler would otherwise pad are avoided by construction (see the notes at the asm blocks).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "mlp_common.h"
#include "pipe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int D = 384, NCH = 12, NKS = 24, NOT = 12, TMR = 128;
constexpr int UNIT = 24 * 1024, NSLOT = 5;      // ring unit = one role's step = 24 fragments of 1 KiB
constexpr int HB_OFF = NSLOT * UNIT;             // fc1 -> fc2 hand-over: [parity 2][pair 4][bf16 fragment 1 KiB | fp32 registers 8..15: 2 KiB]
constexpr int HB_PAR = 4 * 3072;
constexpr int CONST_OFF = HB_OFF + 2 * HB_PAR;    // gamma | beta | b2 | b1[hidden] | tile_s[4] | gamma1 | beta1

#ifndef PSTAMP_SEQ
#define PSTAMP_SEQ 0
#endif
#define PSTAMP(k)                                                                                                                  \
    do {                                                                                                                           \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// The packed image: per tile pass 2 * ntile units (ntile = hidden / 32) of 24 fragments x 1 KiB, lane-major (lane l = 32 h + r: 16
// bytes at l * 16), in the order the steps consume them: unit q of slot k -- slots 0, 1: W1(k); slots 2..ntile-1: W1(k), W2(k-2);
// slots ntile, ntile+1: W2(k-2).
//   W1(k) fragment s (k-step, 0..23)  = W1[32 k + r][32 (s >> 1) + 16 h + 8 (s & 1) + (0..7)]    (the k order of the activations)
//   W2(j) fragment 12 s' + O (output tile O, k-step s') : element e = W2[32 O + r][32 j + 16 s' + 8 (e >> 2) + 4 h + (e & 3)]
//                                                                                  (the hidden order of a GELU'd fc1 tile)
__global__ void mlp_ws_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, int hidden, u32x4* __restrict__ out) {
    const int ntile = hidden / 32, upt = 2 * ntile;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte lane chunk
    if (i >= (int64_t)upt * (UNIT / 16)) return;
    const int q = (int)(i / (UNIT / 16)), o = (int)(i % (UNIT / 16)), frag = o >> 6, lane = o & 63, r = lane & 31, h = lane >> 5;
    // unit q -> (is W1, tile index)
    bool is1;
    int t;
    if (q < 2) {
        is1 = true;
        t = q;
    } else if (q >= upt - 2) {
        is1 = false;
        t = ntile - (upt - q);
    } else {
        is1 = (q & 1) == 0;
        t = is1 ? (q + 2) / 2 : (q - 3) / 2;
    }
    if (is1) {
        out[i] = *(const u32x4*)(w1 + (int64_t)(32 * t + r) * D + 32 * (frag >> 1) + 16 * h + 8 * (frag & 1));
    } else {
        const int O = frag % 12, s2 = frag / 12;
        const bf16_t* row = w2 + (int64_t)(32 * O + r) * hidden + 32 * t + 16 * s2 + 4 * h;
        const u32x2 lo = *(const u32x2*)row, hi = *(const u32x2*)(row + 8);
        out[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
}

// ln_rows_lds of pipe_common.h for a 128-VGPR budget: the same arithmetic in the same order (bitwise the same result), gamma / beta
// fetched per half chunk so that at most 8 registers of them are live beside the 96 of the fragment.
template <int D_, int NCH_>
__device__ __forceinline__ void ln_rows_lds_small(f32x4 (&v)[NCH_][2], uint32_t gaddr, float eps, u32x4 (&out)[NCH_]) {
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH_; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[c][0][e] + v[c][1][e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / D_);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH_; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = v[c][0][e] - mean, b = v[c][1][e] - mean;
            q = __builtin_fmaf(a, a, q);
            q = __builtin_fmaf(b, b, q);
        }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / D_) + eps);
    sfor<0, NCH_>([&](auto C_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        constexpr int c = decltype(C_)::value;
        u32x4 o;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            f32x4 g0, b0;
            const uint32_t ga = gaddr;
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(g0), "=&v"(b0)
                         : "v"(ga), "n"(c * 128 + 16 * hh), "n"(c * 128 + D_ * 4 + 16 * hh));
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((v[c][hh][e] - mean) * rstd, g0[e], b0[e]);
            o[2 * hh] = pack_bf16x2(y[0], y[1]);
            o[2 * hh + 1] = pack_bf16x2(y[2], y[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+a"(o));  // straight to the accumulator file: the arch VGPRs hold the fp32 fragment
        out[c] = o;
    });
}

#ifdef WS_NO_MFMA  // (probe builds: timing without the matrix pipe)
#define MF_OP "; v_mfma_f32_32x32x16_bf16"
#else
#define MF_OP "v_mfma_f32_32x32x16_bf16"
#endif
#define WS_MFMA_AA(acc, a, b) asm volatile(MF_OP " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b))
#define WS_MFMA_AA0(acc, a, b) asm volatile(MF_OP " %0, %1, %2, 0" : "=a"(acc) : "v"(a), "a"(b))
#define WS_MFMA_AV(acc, a, b) asm volatile(MF_OP " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define WS_MFMA_VV(acc, a, b) asm volatile(MF_OP " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define WS_RD(dst, addr, off)                 \
    do {                                      \
        if constexpr (!(DBG & 4)) DSR128(dst, addr, off); \
    } while (0)
#define WS_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
#define WS_DSW128(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")

// IMG / XIN: fragment-blocked activation images (kernels.h) -- IMG: y1 is read and x / xn_out are written as images; XIN: x is read
// as an image.  Weights always come from the packed image p.wpk (format 2).
// DBG (tools/mlp_probe.hip, timing only -- results are garbage): 1 no LDS-DMA in the steps, 2 no GELU arithmetic, 4 no LDS fragment
// reads, 8 no step barriers, 16 no row phases, 32 no MFMAs.
template <bool IMG = false, bool XIN = false, int DBG = 0>
__global__ __launch_bounds__(512, 1) void mlp_ws_kernel(const MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gam = (float*)(smem + CONST_OFF);
    float* bet = gam + D;
    float* b2s = bet + D;
    float* b1s = b2s + D;                  // [hidden]
    int* tile_s = (int*)(b1s + p.hidden);  // [2] tile handed to this workgroup, double-buffered by parity
    float* gam1 = (float*)(tile_s + 4);    // next block's LayerNorm-1 (gamma | beta), if p.xn_out

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave8 >> 2, pw = wave8 & 3;    // role 0 = F1, 1 = F2; pw = the pair's 32-row quarter of the tile
    const int li = lane & 15, g = lane >> 4;        // the row phases' 16-row fragment view: lane (li, g) owns chunks g + 4c
    const int h = lane >> 5, m = (lane >> 4) & 1;   // the MFMA view: lane = 32 h + 16 m + li holds row (fragment m, li), k half h
    const int ntile = p.hidden / 32, upt = 2 * ntile;

    // ---- weight DMA: every wave issues pieces 3 w .. 3 w + 2 of each unit ----
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 2 * p.hidden * D * 2, 0x00020000);
    const uint32_t ilane = (uint32_t)(3 * wave8 * 1024 + lane * 16);
    int ig_slot = 0, ig_off = 0;  // the unit being issued: ring slot, image offset
    // Piece t (0..2) of the wave's three.  An LDS-DMA instruction takes its LDS base from M0, and WRITING M0 is what makes a piece
    // expensive (tools/issue_mix_probe.hip: +36 cycles per piece with a new M0, +2 with the same M0 and the piece selected by the
    // instruction's immediate offset, which is added to the LDS and to the global address alike): a unit's three share one M0.
    auto dma_t = [&](auto T_) __attribute__((always_inline)) {
        constexpr int t = decltype(T_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + ig_slot * UNIT + 3 * wave8 * 1024), 16, ilane, ig_off, t * 1024, 0);
        if constexpr (t == 2) {
            ig_slot = ig_slot == NSLOT - 1 ? 0 : ig_slot + 1;
            ig_off = ig_off + UNIT == upt * UNIT ? 0 : ig_off + UNIT;
        }
    };
    auto dma_unit = [&]() __attribute__((always_inline)) { sfor<0, 3>(dma_t); };
    // the pieces of NU units (0, 1 or 2) spread over a step's 24 MFMA gaps: is gap s one, and which piece
#define WS_DMA_IN_GAP(NU, s)                                                                 \
    do {                                                                                     \
        if constexpr ((NU) > 0 && !(DBG & 1)) {                                              \
            if constexpr ((s) % (8 / (NU)) == 0) dma_t(std::integral_constant<int, ((s) / (8 / (NU))) % 3>{}); \
        }                                                                                    \
    } while (0)

    for (int i = tid; i < D; i += 512) {
        gam[i] = p.ln_w[i];
        bet[i] = p.ln_b[i];
        b2s[i] = p.b2[i];
        if (p.xn_out) {
            gam1[i] = p.ln_next_w[i];
            gam1[D + i] = p.ln_next_b[i];
        }
    }
    for (int i = tid; i < p.hidden; i += 512) b1s[i] = p.b1[i];
    if (tid == 0) tile_s[0] = atomicAdd(p.counter, 1);
    __syncthreads();
    int tile = __builtin_amdgcn_readfirstlane(tile_s[0]);
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memrealtime();

    const uint32_t lbase = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    const uint32_t fbase = lbase + lane * 16;                                            // + slot * UNIT + fragment * 1024
    const uint32_t hbase = lbase + HB_OFF + pw * 3072 + lane * 16;                       // + parity * HB_PAR (+ 1024, 2048: the fp32 halves)
    const uint32_t b1base = (uint32_t)(uintptr_t)(LDS_AS char*)b1s + 16 * h;             // b1[32 t + 8 q + 4 h ..]: + (32 t + 8 q) * 4
    const uint32_t tsbase = (uint32_t)(uintptr_t)(LDS_AS char*)tile_s;
    const uint32_t gbase = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 32 * g;              // (row phases: 16-row fragment view)
    const uint32_t b2base = (uint32_t)(uintptr_t)(LDS_AS char*)b2s + 16 * h;             // b2[32 O + 8 q + 4 h ..]: + (32 O + 8 q) * 4
    const uint32_t g1base = (uint32_t)(uintptr_t)(LDS_AS char*)gam1 + 16 * h;            // next LN-1 gamma (beta: + D * 4)

    // ---- prime the ring: the first NSLOT - 1 units of the stream (step 0 issues the one that fills it) ----
    if (tile < p.ntiles) {
#pragma unroll
        for (int i = 0; i < NSLOT - 1; ++i) dma_unit();
    }

    // ================= role state =================
    u32x4 X[NKS];        // F1: LN2(x + y1) as the fc1 B operand: k-step s, lane half h: k = 32 (s >> 1) + 16 h + 8 (s & 1) + (0..7)
    f32x16 acc1[2];      // F1: fc1 tile accumulators (step parity)
    f32x16 acc2[NOT];    // F2: lane holds its row's output columns 32 O + 8 (reg >> 2) + 4 h + (reg & 3)
    u32x4 wA[4];         // both: rolling weight fragments (fragment s in wA[s & 3], read three MFMAs ahead)
    const float c1v = -1.067757332e-01f;  // GELU: the one coefficient that has to sit in a VGPR (v_fmamk_f32); hipcc re-materialises it

    // ---- F1 row phase: v = x + y1 -> LN2 -> operand fragments (16-row fragment view), then the 32-row B operand ----
    auto prologue = [&](int row0, int nrows) __attribute__((always_inline)) {
        u32x4 af[2][NCH];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            int r = (pw * 2 + mf) * 16 + li;
            r = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
            // image forms: whole fragments only (the launcher guarantees M % 16 == 0); a fragment past the tile's end re-reads
            // fragment 0 of the tile (never stored)
            int fr = (pw * 2 + mf) * 16 < nrows ? (pw * 2 + mf) * 16 : 0;
            // (the second fragment's addresses exist only after the first one is parked: hoisted above its LayerNorm, the loads
            //  of both fragments would be live together and spill -- 128 arch VGPRs per wave here)
            asm volatile("" : "+v"(r), "+s"(fr));
            const float* xr = XIN ? p.x + (int64_t)(row0 + fr) * D + lane * 4 : p.x + (int64_t)(row0 + r) * D + g * 8;
            constexpr int xc_ = XIN ? 512 : 32, xh_ = XIN ? 256 : 4;
            f32x4 v[NCH][2];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                v[c][0] = *(const f32x4*)(xr + c * xc_);
                v[c][1] = *(const f32x4*)(xr + c * xc_ + xh_);
            }
            if (p.y1) {
                // (128 arch VGPRs per wave here: the fp32 fragment takes 96 of them, so y1 comes in three batches of four chunks)
                const bf16_t* yr = IMG ? (const bf16_t*)p.y1 + (int64_t)(row0 + fr) * D + lane * 8 : (const bf16_t*)p.y1 + (int64_t)(row0 + r) * D + g * 8;
                constexpr int yc_ = IMG ? 512 : 32;
#pragma unroll
                for (int cb = 0; cb < NCH; cb += 4) {
                    u32x4 yv[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) yv[c] = *(const u32x4*)(yr + (cb + c) * yc_);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bf16x8 y = __builtin_bit_cast(bf16x8, yv[c]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[cb + c][0][e] += (float)y[e];
                            v[cb + c][1][e] += (float)y[4 + e];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ln_rows_lds_small<D, NCH>(v, gbase, p.ln_eps, af[mf]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // lanes l and l ^ 16 hold each other's missing chunks: one v_permlane16_swap per dword (see mlp32.hip)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            u32x4 e4, o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(af[0][c][e], af[1][c][e], false, false);
                e4[e] = sw[0];
                o4[e] = sw[1];
            }
            X[2 * c] = e4;
            X[2 * c + 1] = o4;
        }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            u32x4& xs = X[s];
            asm volatile("" : "+a"(xs));  // home: the accumulator file (the steps name it with "a" constraints)
        }
    };

    // GELU of two accumulator values in three stages (gelu1 of mlp_common.h on x = acc + bias:  t = min(x^2, 64);
    // q = (c2 t + c1) t + c0;  y = x / (1 + 2^(x q))), the result packed as one bf16x2 word.  The two elements' chains alternate, so
    // no instruction reads the result of the one just before it (a transcendental's result needs one instruction in between, and
    // hipcc does not pad inside asm).
#define WS_GELU_ST0(a0, b0, a1, b1)                                                                      \
    asm volatile(                                                                                         \
        "v_add_f32 %0, %4, %5\n\tv_add_f32 %1, %6, %7\n\t"                                                \
        "v_mul_f32 %2, %0, %0\n\tv_mul_f32 %3, %1, %1\n\t"                                                \
        "v_min_f32 %2, 0x42800000, %2\n\tv_min_f32 %3, 0x42800000, %3"                                    \
        : "=&v"(xa), "=&v"(xb), "=&v"(ta), "=&v"(tb)                                                      \
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1))
#define WS_GELU_ST0_NB()  /* xa, xb already hold acc + bias */                                           \
    asm volatile(                                                                                         \
        "v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3\n\t"                                                \
        "v_min_f32 %0, 0x42800000, %0\n\tv_min_f32 %1, 0x42800000, %1"                                    \
        : "=&v"(ta), "=&v"(tb)                                                                            \
        : "v"(xa), "v"(xb))
#define WS_GELU_ST1()                                                                                     \
    asm volatile(                                                                                         \
        "v_fmamk_f32 %0, %2, 0x3a84f112, %6\n\tv_fmamk_f32 %1, %3, 0x3a84f112, %6\n\t"                    \
        "v_fmaak_f32 %0, %0, %2, 0xc0134592\n\tv_fmaak_f32 %1, %1, %3, 0xc0134592\n\t"                    \
        "v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %5, %1\n\t"                                                \
        "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1"                                                            \
        : "=&v"(qa), "=&v"(qb)                                                                            \
        : "v"(ta), "v"(tb), "v"(xa), "v"(xb), "v"(c1v))
#define WS_GELU_ST2(wo)                                                                                   \
    asm volatile(                                                                                         \
        "v_add_f32 %1, 1.0, %1\n\tv_add_f32 %2, 1.0, %2\n\t"                                              \
        "v_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\t"                                                        \
        "v_mul_f32 %3, %3, %1\n\tv_mul_f32 %4, %4, %2\n\t"                                                \
        "v_cvt_pk_bf16_f32 %0, %3, %4"                                                                    \
        : "=v"(wo), "+v"(qa), "+v"(qb), "+v"(xa), "+v"(xb))

    // ---- F1 step: MF: the 24 MFMAs of hidden tile kt into acc1[PAR]; GE: tile kt - 1 (acc1[PAR ^ 1]) leaves for F2 through
    // hbuf[PAR ^ 1]: accumulator registers 8..15 + bias as fp32 (F2 finishes them itself), registers 0..7 GELU'd here (4 units,
    // one stage every other MFMA gap) as the fc2 operand fragment of k-step 0.  slot = ring slot of W1(kt)
    auto f1_step = [&](auto PAR_, auto MF_, auto GE_, auto NU_, int kt, int slot) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, NU = decltype(NU_)::value;
        constexpr bool MF = decltype(MF_)::value != 0, GE = decltype(GE_)::value != 0;
        const uint32_t sa = fbase + slot * UNIT, hw = hbase + (PAR ^ 1) * HB_PAR;
        f32x4 bq[4];  // bias of tile kt - 1: bq[q][e] = b1[32 (kt-1) + 8 q + 4 h + e] = the lane's accumulator register 4 q + e
        uint32_t w4[4];
        float xa = 0.f, xb = 0.f, ta = 0.f, tb = 0.f, qa = 0.f, qb = 0.f;
        if constexpr (GE) {
            const uint32_t ba = b1base + (kt - 1) * 128;
            f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
            DSR128(q2, ba, 64);
            DSR128(q3, ba, 96);
            DSR128(q0, ba, 0);
            DSR128(q1, ba, 32);
        }
        if constexpr (MF) {
            u32x4 &d0 = wA[0], &d1 = wA[1], &d2 = wA[2];
            WS_RD(d0, sa, 0);
            WS_RD(d1, sa, 1024);
            WS_RD(d2, sa, 2048);
        } else if constexpr (GE) {
            // (no MFMA in this step: the previous step's last MFMA needs ~11 issue slots before compiled code may read its result)
            asm volatile("s_nop 7\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        }
        sfor<0, NKS>([&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            if constexpr (GE && s == 1) {
                // registers 8..15 of tile kt - 1 leave with their bias added, still fp32 (F2 has the issue slots for their GELU)
                const f32x16& pa = acc1[PAR ^ 1];
                f32x4 r0, r1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r0[e] = pa[8 + e] + bq[2][e];
                    r1[e] = pa[12 + e] + bq[3][e];
                }
                const uint32_t hwl = hw;
                WS_DSW128(hwl, r0, 1024);
                WS_DSW128(hwl, r1, 2048);
            }
            if constexpr (MF) {
                u32x4 &fa = wA[s & 3], &xs = X[s];
                f32x16& ac = acc1[PAR];
                WS_LGKM((NKS - 1 - s) < 2 ? (NKS - 1 - s) : 2);
                if constexpr (s == 0) {
                    WS_MFMA_AA0(ac, fa, xs);
                } else {
                    WS_MFMA_AA(ac, fa, xs);
                }
                if constexpr (s + 3 < NKS) {
                    u32x4& fn = wA[(s + 3) & 3];
                    WS_RD(fn, sa, (s + 3) * 1024);
                }
            }
            WS_DMA_IN_GAP(NU, s);
            if constexpr (GE && s == 0) {
                // hipcc reads acc1[PAR ^ 1] from the accumulator file wherever it likes AFTER this statement -- not before: it does
                // not know that an MFMA (inside asm) wrote it at the end of the previous step and needs ~11 issue slots to land
                f32x16& pa = acc1[PAR ^ 1];
                asm volatile("s_nop 3" : "+a"(pa));
            }
            if constexpr (GE && (s & 1) == 0) {
                constexpr int u = s / 6, st = (s % 6) / 2;  // unit u: accumulator registers 2 u, 2 u + 1
                if constexpr (DBG & 2) {
                    if constexpr (st == 2) w4[u] = __builtin_bit_cast(uint32_t, acc1[PAR ^ 1][2 * u]);
                } else if constexpr (st == 0) {
                    const float a0 = acc1[PAR ^ 1][2 * u], a1 = acc1[PAR ^ 1][2 * u + 1];
                    const float b0 = bq[u >> 1][2 * (u & 1)], b1 = bq[u >> 1][2 * (u & 1) + 1];
                    WS_GELU_ST0(a0, b0, a1, b1);
                } else if constexpr (st == 1) {
                    WS_GELU_ST1();
                } else {
                    uint32_t& wo = w4[u];
                    WS_GELU_ST2(wo);
                }
                if constexpr (u == 3 && st == 2) {
                    const u32x4 f0 = {w4[0], w4[1], w4[2], w4[3]};
                    const uint32_t hwl = hw;  // (a name used only as an asm operand is not captured by a generic lambda)
                    WS_DSW128(hwl, f0, 0);
                }
            }
        });
        if constexpr (GE) WS_LGKM(0);  // what F2 reads next step is in LDS before the step's barrier
    };

    // ---- F2 step: the fc2 MFMAs of one hidden tile.  From hbuf[PARH]: the operand fragment of k-step 0 and the fp32 accumulator
    // registers 8..15, which it GELUs itself (4 units, one stage per gap of the first twelve MFMAs) into the fragment of k-step 1;
    // MFMA i < 12: output tile i, k-step 0; i >= 12: output tile i - 12, k-step 1 (the W2 unit lists its fragments in that order).
    // Weights from ring slot `slot`; issues its LDS-DMA pieces of NU units (3 per unit and wave) in its MFMA gaps.  jt = hidden tile.
    auto f2_step = [&](auto PARH_, auto NU_, int jt, int slot) __attribute__((always_inline)) {
        constexpr int PARH = decltype(PARH_)::value, NU = decltype(NU_)::value;
        const uint32_t sa = fbase + slot * UNIT, ha = hbase + PARH * HB_PAR;
        (void)jt;
        u32x4 hB0, hB1;
        f32x4 rr[2];  // accumulator registers 8..15 of the tile, bias added
        uint32_t w4[4];
        float xa = 0.f, xb = 0.f, ta = 0.f, tb = 0.f, qa = 0.f, qb = 0.f;
        {
            f32x4 &r0 = rr[0], &r1 = rr[1];
            DSR128(hB0, ha, 0);
            DSR128(r0, ha, 1024);
            DSR128(r1, ha, 2048);
            u32x4 &d0 = wA[0], &d1 = wA[1], &d2 = wA[2];
            WS_RD(d0, sa, 0);
            WS_RD(d1, sa, 1024);
            WS_RD(d2, sa, 2048);
        }
        sfor<0, NKS>([&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value, O = s % 12;
            u32x4& fa = wA[s & 3];
            f32x16& ac = acc2[O];
            WS_LGKM((NKS - 1 - s) < 2 ? (NKS - 1 - s) : 2);
            if constexpr (s == 12) {
                hB1 = u32x4{w4[0], w4[1], w4[2], w4[3]};
                asm volatile("s_nop 1" : "+v"(hB1));  // (VALU result -> MFMA operand)
            }
            u32x4& hb = s < 12 ? hB0 : hB1;
            if constexpr (O < 8) {
                WS_MFMA_AV(ac, fa, hb);
            } else {
                WS_MFMA_VV(ac, fa, hb);
            }
            if constexpr (s + 3 < NKS) {
                u32x4& fn = wA[(s + 3) & 3];
                WS_RD(fn, sa, (s + 3) * 1024);
            }
            WS_DMA_IN_GAP(NU, s);
            if constexpr (s < 12) {
                constexpr int u = s / 3, st = s % 3;  // unit u: accumulator registers 8 + 2 u, 9 + 2 u
                if constexpr (DBG & 2) {
                    if constexpr (st == 2) w4[u] = __builtin_bit_cast(uint32_t, rr[u >> 1][2 * (u & 1)]);
                } else if constexpr (st == 0) {
                    xa = rr[u >> 1][2 * (u & 1)];
                    xb = rr[u >> 1][2 * (u & 1) + 1];
                    WS_GELU_ST0_NB();
                } else if constexpr (st == 1) {
                    WS_GELU_ST1();
                } else {
                    uint32_t& wo = w4[u];
                    WS_GELU_ST2(wo);
                }
            }
        });
    };

    // ---- F2 row phase: x <- x + y1 + acc2 + b2, then the next block's LayerNorm-1 (see mlp32.hip for the mapping) ----
    auto epilogue = [&](int row0, int nrows) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        const int r = pw * 32 + m * 16 + li;
        // rows past the end of a tail tile: their lanes sit the row phase out (ONE branch: the two lanes of a row, l and l ^ 32,
        // are live together, and nothing below exchanges data between rows)
        if (r >= nrows) return;
        const int fr = pw * 32 + m * 16;
        const int64_t rb = (int64_t)(row0 + r) * D + 4 * h, fb = (int64_t)(row0 + fr) * D;
        const float* xl = XIN ? p.x + fb + 256 * h + 4 * li : p.x + rb;
        float* xs = IMG ? p.x + fb + 256 * h + 4 * li : p.x + rb;
        const bf16_t* yr = IMG ? (const bf16_t*)p.y1 + fb + 8 * li + 4 * h : (const bf16_t*)p.y1 + rb;
        constexpr int xlo_ = XIN ? 512 : 32, xlq_ = XIN ? 64 : 8, xso_ = IMG ? 512 : 32, xsq_ = IMG ? 64 : 8, yo_ = IMG ? 512 : 32, yq_ = IMG ? 128 : 8;
        float rsum = 0.f;
        if constexpr (XIN != IMG) {
            // converting in place (row-major in, image out: the first block of a forward) a lane's stores land where OTHER lanes'
            // loads read: move every old value of the tile into the accumulators first
#pragma unroll
            for (int O = 0; O < NOT; ++O)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 xv = *(const f32x4*)(xl + xlo_ * O + xlq_ * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc2[O][4 * q + e] += xv[e];
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a fragment's 16 rows belong to ONE wave: no other wave reads here)
        }
        sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
            constexpr int O = decltype(O_)::value;
            f32x4 xv[4];
            u32x2 yv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (XIN == IMG) xv[q] = *(const f32x4*)(xl + xlo_ * O + xlq_ * q);
                yv[q] = p.y1 ? *(const u32x2*)(yr + yo_ * O + yq_ * q) : u32x2{0u, 0u};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bb;  // b2[32 O + 8 q + 4 h ..]
                {
                    const uint32_t ba = b2base;
                    asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(bb) : "v"(ba), "n"(O * 128 + q * 32));
                }
                const bf16x4 y = __builtin_bit_cast(bf16x4, yv[q]);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (XIN == IMG)
                        v[e] = ((acc2[O][4 * q + e] + bb[e]) + xv[q][e]) + (float)y[e];
                    else
                        v[e] = (acc2[O][4 * q + e] + bb[e]) + (float)y[e];
                }
                *(f32x4*)(xs + xso_ * O + xsq_ * q) = v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc2[O][4 * q + e] = v[e];
                    rsum += v[e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // (64 + 40 arch VGPRs are in use here: loads hoisted from the next tile would spill)
        });
        if (p.xn_out) {
            rsum += __shfl_xor(rsum, 32, 64);
            const float mean = rsum * (1.0f / D);
            float qs = 0.f;
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {  // (one tile at a time: 128 arch VGPRs here)
#pragma clang fp contract(off)
                constexpr int O = decltype(O_)::value;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float a = acc2[O][e] - mean;
                    qs = __builtin_fmaf(a, a, qs);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            qs += __shfl_xor(qs, 32, 64);
            const float rstd = 1.0f / sqrtf(qs * (1.0f / D) + p.ln_eps);
            bf16_t* nr = IMG ? (bf16_t*)p.xn_out + fb + 8 * li + 4 * h : (bf16_t*)p.xn_out + rb;
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int O = decltype(O_)::value;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 gq, bqv;  // next LayerNorm-1: gamma, beta [32 O + 8 q + 4 h ..]
                    {
                        const uint32_t ga = g1base;
                        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(gq), "=&v"(bqv)
                                     : "v"(ga), "n"(O * 128 + q * 32), "n"(D * 4 + O * 128 + q * 32));
                    }
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc2[O][4 * q + e] - mean) * rstd, gq[e], bqv[e]);
                    u32x2 o2;
                    o2[0] = pack_bf16x2(y[0], y[1]);
                    o2[1] = pack_bf16x2(y[2], y[3]);
                    *(u32x2*)(nr + yo_ * O + yq_ * q) = o2;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    };

    auto tile_rows = [&](int t, int& row0, int& nrows) __attribute__((always_inline)) {
        // tiles [0, full_tiles): 128 rows each; then 16-row tail tiles (only pair 0 / fragment 0 has rows)
        if (t < p.full_tiles) {
            row0 = t * TMR;
            nrows = TMR;
        } else {
            row0 = p.full_tiles * TMR + (t - p.full_tiles) * 16;
            nrows = 16;
        }
        nrows = (p.M - row0) < nrows ? (p.M - row0) : nrows;
    };

    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 2> I2;

    int gfirst = 0;  // ring slot of the first unit of the current step
    unsigned long long barw = 0, vmw = 0, tstart = __builtin_amdgcn_s_memtime();  // (debug stamps: cycles spent in the step barriers / DMA waits)
    // A step ends with the workgroup's barrier; the ring advances by the n units the step consumed.  Every wave first waits until its pieces
    // of the NEXT step's units have landed: all but the 3 newest, which belong to the step after.
    // The two roles run their own tile loops (so that hipcc sees each role's registers live only in its own loop); both execute the
    // same number of barriers per tile: ntile + 2.
    auto ring = [&](int o) __attribute__((always_inline)) { return gfirst + o >= NSLOT ? gfirst + o - NSLOT : gfirst + o; };
#define WS_ADVANCE(n)                                         \
    do {                                                      \
        gfirst += (n);                                        \
        gfirst = gfirst >= NSLOT ? gfirst - NSLOT : gfirst;   \
    } while (0)
#define WS_BAR()                                                      \
    do {                                                              \
        if constexpr (!(DBG & 8)) {                                   \
            if (HIPT_STAMPS_ON(p.stamps)) {                           \
                const unsigned long long tb0 = __builtin_amdgcn_s_memtime(); \
                __builtin_amdgcn_s_barrier();                         \
                barw += __builtin_amdgcn_s_memtime() - tb0;           \
            } else {                                                  \
                __builtin_amdgcn_s_barrier();                         \
            }                                                         \
        }                                                             \
    } while (0)
#define WS_END_F1(n)                     \
    do {                                 \
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); \
        WS_BAR();    \
        WS_ADVANCE(n);                   \
    } while (0)
#define WS_END_F2(n)                                          \
    do {                                                      \
        if (HIPT_STAMPS_ON(p.stamps)) {                       \
            const unsigned long long tv0 = __builtin_amdgcn_s_memtime(); \
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); \
            vmw += __builtin_amdgcn_s_memtime() - tv0;        \
        } else {                                              \
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); \
        }                                                     \
        WS_BAR();                         \
        WS_ADVANCE(n);                                        \
    } while (0)

    int row0 = 0, nrows = 0;
    if (tile < p.ntiles) tile_rows(tile, row0, nrows);

    if (role == 0) {
        // ======================= F1 =======================
        if (tile < p.ntiles) prologue(row0, nrows);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int seq = 0; tile < p.ntiles; ++seq) {
            PSTAMP(0);
            if (tid == 0) {  // next tile: fetched now, read many barriers later
                const int nt = atomicAdd(p.counter, 1);
                asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt) : "memory");
            }
            f1_step(I0{}, I1{}, I0{}, I1{}, 0, gfirst);  // steps 0, 1: fc1 tiles 0, 1 (F2 only feeds the ring)
            WS_END_F1(1);
            f1_step(I1{}, I1{}, I1{}, I1{}, 1, gfirst);
            WS_END_F1(1);
            f1_step(I0{}, I1{}, I1{}, I1{}, 2, gfirst);
            WS_END_F1(2);
            PSTAMP(2);
            for (int kk = 3; kk + 1 < ntile; kk += 2) {
                f1_step(I1{}, I1{}, I1{}, I2{}, kk, gfirst);
                WS_END_F1(2);
                f1_step(I0{}, I1{}, I1{}, I2{}, kk + 1, gfirst);
                WS_END_F1(2);
            }
            f1_step(I1{}, I1{}, I1{}, I2{}, ntile - 1, gfirst);
            WS_END_F1(2);
            f1_step(I0{}, I0{}, I1{}, I2{}, ntile, gfirst);  // step ntile: GELU of the last fc1 tile
            WS_END_F1(1);
            PSTAMP(3);
            // step ntile + 1: the next tile's row phase (beside F2's last step + row phase of this tile)
            int nt;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nt) : "v"(tsbase + 4 * ((seq + 1) & 1)) : "memory");
            tile = __builtin_amdgcn_readfirstlane(nt);
            if constexpr (!(DBG & 1)) dma_unit();
            if (tile < p.ntiles) {
                tile_rows(tile, row0, nrows);
                if constexpr (!(DBG & 16)) prologue(row0, nrows);
            }
            WS_END_F1(1);
            PSTAMP(4);
        }
    } else {
        // ======================= F2 =======================
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring is primed
        __builtin_amdgcn_s_barrier();
        for (int seq = 0; tile < p.ntiles; ++seq) {
#pragma unroll
            for (int o = 0; o < NOT; ++o)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[o][e] = 0.f;
            if constexpr (!(DBG & 1)) dma_unit();  // steps 0, 1: nothing to multiply yet; one unit per step keeps the ring full
            WS_END_F2(1);
            if constexpr (!(DBG & 1)) dma_unit();
            WS_END_F2(1);
            f2_step(I0{}, I1{}, 0, ring(1));  // step 2: fc2 of tile 0
            WS_END_F2(2);
            for (int kk = 3; kk + 1 < ntile; kk += 2) {
                f2_step(I1{}, I2{}, kk - 2, ring(1));
                WS_END_F2(2);
                f2_step(I0{}, I2{}, kk - 1, ring(1));
                WS_END_F2(2);
            }
            f2_step(I1{}, I2{}, ntile - 3, ring(1));
            WS_END_F2(2);
            f2_step(I0{}, I2{}, ntile - 2, gfirst);  // step ntile: fc2 of tile ntile - 2
            WS_END_F2(1);
            // step ntile + 1: fc2 of the last tile, then this tile's row phase
            int nt;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nt) : "v"(tsbase + 4 * ((seq + 1) & 1)) : "memory");
            f2_step(I1{}, I1{}, ntile - 1, gfirst);
            // the accumulators are read by compiled code from here on: hipcc does not know an MFMA has just written them
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {  // (... and may not read them before the nops either)
                constexpr int O = decltype(O_)::value;
                f32x16& ac = acc2[O];
                if constexpr (O < 8) {
                    asm volatile("" : "+a"(ac));
                } else {
                    asm volatile("" : "+v"(ac));
                }
            });
            if constexpr (!(DBG & 16)) epilogue(row0, nrows);
            // (no counted wait: what the next two steps read was issued two steps ago, and this wave's row loads came back after it)
            __builtin_amdgcn_s_barrier();
            WS_ADVANCE(1);
            tile = __builtin_amdgcn_readfirstlane(nt);
            if (tile < p.ntiles) tile_rows(tile, row0, nrows);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: pieces of a pass that never runs)
    if (HIPT_STAMPS_ON(p.stamps) && lane == 0 && pw == 0) {
        p.stamps[(size_t)blockIdx.x * 16 + 12 + role] = barw;
        p.stamps[(size_t)blockIdx.x * 16 + 14 + role] = role == 0 ? __builtin_amdgcn_s_memtime() - tstart : vmw;
    }
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

bool hipt_mlp_ws_supported(int dtype, int D_, int hidden) {
    return dtype == HIPT_BF16 && D_ == 384 && hidden % 64 == 0 && hidden >= 256 && hidden <= 2048;
}

int hipt_mlp_ws_pack_launch(const void* w1, const void* w2, int D_, int hidden, void* packed, hipStream_t st) {
    if (!(D_ == 384 && hidden % 64 == 0 && hidden >= 256 && hidden <= 2048)) {
        hipt_set_error("mlp_ws pack: unsupported D=%d hidden=%d", D_, hidden);
        return HIPT_E_UNSUPPORTED;
    }
    const int64_t chunks = (int64_t)(hidden / 32) * 2 * (UNIT / 16);
    hipLaunchKernelGGL(mlp_ws_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w1, (const bf16_t*)w2, hidden, (u32x4*)packed);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

template <int DBG>
int hipt_mlp_ws_launch_dbg(const MlpParams& p_in, hipStream_t st) {
    MlpParams p = p_in;
    const int lds = CONST_OFF + (3 * D + p.hidden) * 4 + 16 + 2 * D * 4;
    if (!p.wpk || p.wpk_fmt != 2 || (p.img & 2 && !(p.img & 1)) || (p.img && p.M % 16 != 0)) {
        hipt_set_error("mlp_ws: needs its packed weights; activation images need M %% 16 == 0 and img in {0, 1, 3} (img=%d, M=%d)", p.img, p.M);
        return HIPT_E_BADARG;
    }
    auto k = p.img == 3 ? mlp_ws_kernel<true, true, DBG> : p.img == 1 ? mlp_ws_kernel<true, false, DBG> : mlp_ws_kernel<false, false, DBG>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)mlp_ws_kernel<true, true, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp_ws_kernel<true, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp_ws_kernel<false, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(mlp_ws) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("mlp_ws: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    const int ncu = once.ncu[dev];
    // whole rounds of #CU workgroups take 128 rows each; a last partial round that would be less than an eighth full is cut into
    // 16-row tiles (one active 16-row fragment each)
    const int tiles = (p.M + TMR - 1) / TMR;
    const int rem = tiles % ncu;
    const int tail_tiles = (tiles > ncu && rem > 0 && rem <= ncu / 8) ? rem : 0;
    p.full_tiles = tiles - tail_tiles;
    const int tail_rows = p.M - p.full_tiles * TMR;
    p.ntiles = p.full_tiles + (tail_rows > 0 ? (tail_rows + 15) / 16 : 0);
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    p.stagger = 0;
    if (hipMemsetAsync(p.counter, 0, sizeof(int), st) != hipSuccess) {
        hipt_set_error("mlp_ws: hipMemsetAsync(counter) failed");
        return HIPT_E_LAUNCH;
    }
#ifdef HIPT_DEBUG_STAMPS  // diagnostic builds only (make DEBUG_STAMPS=1): the release library never allocates or synchronises
    static const bool want_stamps = getenv("HIPT_SEQGEMM_STAMPS") != nullptr;
    static unsigned long long* dbuf = nullptr;
    if (want_stamps) {
        if (!dbuf) (void)hipMalloc(&dbuf, 4096 * 16 * sizeof(unsigned long long));
        (void)hipMemsetAsync(dbuf, 0, 4096 * 16 * sizeof(unsigned long long), st);
        p.stamps = dbuf;
    }
#endif
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, p);
    HIPT_CHECK_LAUNCH();
    // (a caller that zeroes the queue once for a chain of launches -- MlpParams::counter_zeroed -- gets it back zero: this kernel's queue does not reset itself)
    if (p.counter_zeroed) (void)hipMemsetAsync(p.counter, 0, sizeof(int), st);
#ifdef HIPT_DEBUG_STAMPS
    if (want_stamps && grid <= 4096) {
        static unsigned long long h[4096 * 16];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbuf, (size_t)grid * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t4 = 0;
        for (int b = 0; b < grid; ++b) {
            if (h[b * 16 + 11] < t0) t0 = h[b * 16 + 11];
            if (h[b * 16 + 10] > t4) t4 = h[b * 16 + 10];
        }
        double first3 = 0, steps = 0, rows = 0, bw1 = 0, bw2 = 0, tot = 0, vw = 0;
        for (int b = 0; b < grid; ++b) {
            bw1 += (double)h[b * 16 + 12] / grid;
            bw2 += (double)h[b * 16 + 13] / grid;
            tot += (double)h[b * 16 + 14] / grid;
            vw += (double)h[b * 16 + 15] / grid;
            first3 += (double)(h[b * 16 + 2] - h[b * 16 + 0]) * 0.01 / grid;
            steps += (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.01 / grid;
            rows += (double)(h[b * 16 + 4] - h[b * 16 + 3]) * 0.01 / grid;
        }
        fprintf(stderr, "[mlp_ws hidden=%d grid=%d tiles=%d(+%d)] total %.1f us | tile %d of each workgroup: steps 0-2 %.1f, steps 3-%d %.1f, row step %.1f\n",
                p.hidden, grid, p.full_tiles, p.ntiles - p.full_tiles, (double)(t4 - t0) * 0.01, PSTAMP_SEQ, first3, p.hidden / 32, steps, rows);
        fprintf(stderr, "   per workgroup: %.0f kcycles in all; in the step barriers: F1 wave 0 %.0f k, F2 wave 4 %.0f k; F2 wave 4 waiting for its DMA pieces %.0f k\n", tot * 1e-3, bw1 * 1e-3,
                bw2 * 1e-3, vw * 1e-3);
    }
#endif
    return HIPT_OK;
}

int hipt_mlp_ws_launch(const MlpParams& p, hipStream_t st) { return hipt_mlp_ws_launch_dbg<0>(p, st); }
