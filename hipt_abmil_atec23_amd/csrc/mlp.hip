// FUSED MLP SUB-BLOCK (bf16 mode):   x <- x + y1 + fc2( GELU( fc1( LN2(x + y1) ) ) )
//   (Block.forward second half, HIPT_4K/vision_transformer.py:151 with Mlp.forward :98-104; y1 is the
//    attention-branch output the proj kernel left in bf16, i.e. the first residual add :150 is folded in.)
//
// The reference materialises the [M, 4D] hidden tensor (202 MB in bf16 for one region) and re-reads it;
// here it never leaves the chip.  One 4-wave workgroup owns 128 rows (8 MFMA row fragments, 2 per wave)
// and keeps, at one wave per SIMD:
//     af   LN2(x+y1) as operand fragments for fc1           2 x D/8 chunks      (96 VGPRs at D=384)
//     acc1 one 128-wide hidden chunk of fc1                  2 x 8 x 4           (64)
//     hf   GELU(acc1) re-packed IN REGISTERS as the operand of fc2 (accumulator-as-operand: the lane
//          that owns 4 consecutive hidden units of a row after fc1 owns exactly those K slots in fc2;
//          the W2 fragments are read with the matching K permutation)           (32)
//     acc2 the [128, D] output of fc2                        2 x D/16 x 4        (192)
// Only weights stream: per hidden chunk 6 slabs of W1 and 6 of W2 (16 KiB each) through the 8-slot
// LDS-DMA ring shared by the 4 waves.  HBM traffic per call: x and y1 read twice (prologue, epilogue),
// x written once = 403 MB per region instead of 807 MB (LN2 + fc1 + fc2 as separate kernels).
// Row tiling: 65792 rows = 514 tiles of 128; the 2 tiles beyond two full rounds of 256 CUs are cut into
// 16-row workgroups (one active fragment) so the tail costs half a round instead of a whole one.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "mlp_common.h"

namespace {

constexpr int TMR = 128;               // rows per full workgroup
constexpr int SLAB_BYTES = 128 * 128;  // 128 weight rows x 64 bf16
constexpr int NSLOT = 8;

template <int N> __device__ __forceinline__ void wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int U> __device__ __forceinline__ void wait_units(int units) {  // 4*U DMA instructions per unit per wave
    if constexpr (U == 2) {
        switch (units) {
            case 0: wait_vm_n<0>(); break;
            case 1: wait_vm_n<8>(); break;
            default: wait_vm_n<16>(); break;
        }
    } else {
        switch (units) {
            case 0: wait_vm_n<0>(); break;
            case 1: wait_vm_n<4>(); break;
            case 2: wait_vm_n<8>(); break;
            case 3: wait_vm_n<12>(); break;
            case 4: wait_vm_n<16>(); break;
            case 5: wait_vm_n<20>(); break;
            default: wait_vm_n<24>(); break;
        }
    }
}

#define MSTAMP(k)                                                                                   \
    do {                                                                                            \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == 0) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// KS = D / 64 (6: ViT-256, 3: ViT-4K).  DBG (tools/mlp_probe.hip only; the library instantiates 0) removes
// one ingredient at a time to price it: 1 = no weight DMA / waits inside the loop, 2 = GELU replaced by a
// plain pack, 4 = no LDS fragment reads / MFMAs.
//
// Persistent: gridDim.x <= #CUs workgroups pull row tiles from an atomic counter (p.counter, zeroed by the
// launcher).  The weight stream does not depend on the tile, so the ring runs CONTINUOUSLY across tiles
// (the first slabs of the next pass are in flight during a tile's epilogue and the next tile's loads).
// The row loads / stores of a tile are pure HBM time (786 KB per tile, ~5.6 TB/s when all CUs do it at
// once): workgroup groups start p.stagger ticks apart so that some CUs compute while others move rows.
template <int KS, int DBG = 0>
__global__ __launch_bounds__(256, 1) void mlp_kernel(const MlpParams p) {
    constexpr int D = KS * 64;
    constexpr int NCH = KS * 2;          // A chunks per lane per row fragment
    constexpr int NF2 = D / 16;          // fc2 output column fragments (24 / 12)
    constexpr int NG = (D + 127) / 128;  // fc2 weight slabs (128 output rows each) per hidden half
    constexpr int SPC = KS + 2 * NG;     // slabs per hidden chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gam = (float*)(smem + NSLOT * SLAB_BYTES);
    float* bet = gam + D;
    float* b2s = bet + D;
    float* b1s = b2s + D;  // [hidden]
    int* tile_s = (int*)(b1s + p.hidden);  // [2] tile handed to this workgroup, double-buffered by parity

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int nchunk = p.hidden / 128;
    const int nslab = nchunk * SPC;

    // ---- weight slab DMA: per-lane base pointers are fixed; a slab only adds wave-uniform offsets ----
    // U consecutive slabs form one ring unit (one barrier + one counted wait per unit): U = 2 when the
    // per-chunk slab count is even (D = 384: 12), else 1.
    constexpr int U = (SPC % 2 == 0) ? 2 : 1;
    constexpr int NUS = NSLOT / U;  // ring slots in units
    const bf16_t* W1 = (const bf16_t*)p.w1;
    const bf16_t* W2 = (const bf16_t*)p.w2;
    // instruction q of a wave fills rows (q*4 + wave)*8 + (lane>>3): +32 rows per q, and the swizzled
    // chunk does not depend on q, so ONE per-lane base per matrix is enough (q adds a uniform stride)
    const int r0 = wave * 8 + (lane >> 3);
    const int ch0 = (lane & 7) ^ ((r0 >> 1) & 7);
    const bf16_t* w1b = W1 + (int64_t)r0 * D + ch0 * 8;
    const bf16_t* w2b = W2 + (int64_t)r0 * p.hidden + ch0 * 8;
    const int nunit = nslab / U;
    // continuous stream: unit u of the NEXT pass goes to the ring slot after unit nunit-1 of this one,
    // which is slot u again only if a pass is a whole number of ring turns
    const bool cont = (nunit % NUS) == 0;
    auto issue_unit = [&](int u) {
#pragma unroll
        for (int h = 0; h < U; ++h) {
            const int i = u * U + h;
            const int c = i / SPC, j = i % SPC;
            char* dst = smem + ((u % NUS) * U + h) * SLAB_BYTES;
            if (j < KS) {  // fc1: hidden rows [128c, +128) x k [64j, +64)
                const int64_t off = (int64_t)c * 128 * D + j * 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) glds16(w1b + off + q * 32 * D, dst + (q * 4 + wave) * 1024);
            } else {       // fc2: output rows [128ng, +128) x hidden [128c + 64kh, +64)
                const int kh = (j - KS) / NG, ng = (j - KS) % NG;
                const int64_t off = (int64_t)ng * 128 * p.hidden + c * 128 + kh * 64;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16_t* src = w2b + off + (int64_t)q * 32 * p.hidden;
                    if constexpr (D % 128 != 0) {  // last row group of a D = 192 model: clamp rows >= D
                        const int r = (q * 4 + wave) * 8 + (lane >> 3);
                        if (ng * 128 + r >= D) src = W2 + (int64_t)(D - 1) * p.hidden + c * 128 + kh * 64;
                    }
                    glds16(src, dst + (q * 4 + wave) * 1024);
                }
            }
        }
    };
    const int pre = nunit < NUS - 1 ? nunit : NUS - 1;

    for (int i = tid; i < D; i += 256) {
        gam[i] = p.ln_w[i];
        bet[i] = p.ln_b[i];
        b2s[i] = p.b2[i];
    }
    for (int i = tid; i < p.hidden; i += 256) b1s[i] = p.b1[i];
    if (tid == 0) tile_s[0] = atomicAdd(p.counter, 1);
    __syncthreads();
    int tile = __builtin_amdgcn_readfirstlane(tile_s[0]);
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memrealtime();
    if (p.stagger > 0) {  // start groups apart: (block / 8) & 3 mixes the groups inside every XCD
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long wait = (unsigned long long)(((blockIdx.x >> 3) & 3) * p.stagger);
        while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(64);
    }

    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = li * 128 + (((g + 4 * ks) ^ ((lane >> 1) & 7)) << 4);
    // fc2 weight fragment: two 8-byte pieces per lane, hidden offsets 32fl + 4g + (0..3) and + 16, so that
    // the K slots match how GELU(acc1) is packed below.  Byte offset inside the 128-byte row: 64fl + 8g (+32).
    int f2off[2][2];
#pragma unroll
    for (int fl = 0; fl < 2; ++fl)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int byte = 64 * fl + 32 * h + 8 * g;
            f2off[fl][h] = li * 128 + ((((byte >> 4)) ^ ((lane >> 1) & 7)) << 4) + (byte & 8);
        }
    const uint32_t lbase = (uint32_t)(uintptr_t)(LDS_AS char*)smem;

    for (int seq = 0; tile < p.ntiles; ++seq) {
        // tiles [0, full_tiles): 128 rows each; then 16-row tail tiles (only wave 0 / fragment 0 has rows)
        int row0, nrows;
        if (tile < p.full_tiles) {
            row0 = tile * TMR;
            nrows = TMR;
        } else {
            row0 = p.full_tiles * TMR + (tile - p.full_tiles) * 16;
            nrows = 16;
        }
        nrows = (p.M - row0) < nrows ? (p.M - row0) : nrows;
        // fragments this wave really owns (a tail tile has one): lets idle waves skip the matrix work
        const int my_frags = nrows > wave * 32 + 16 ? 2 : (nrows > wave * 32 ? 1 : 0);
        MSTAMP(0);
        if (seq == 0 || !cont)
            for (int u = 0; u < pre; ++u) issue_unit(u);
        if (tid == 0) tile_s[(seq + 1) & 1] = atomicAdd(p.counter, 1);  // read after this tile's barriers
        MSTAMP(1);

        // ---- activations: v = x + y1 -> LN2 -> operand fragments ----
        u32x4 af[2][NCH];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            int r = (wave * 2 + mf) * 16 + li;
            r = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
            const float* xr = p.x + (int64_t)(row0 + r) * D;
            f32x4 v[NCH][2];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                v[c][0] = *(const f32x4*)(xr + (g + 4 * c) * 8);
                v[c][1] = *(const f32x4*)(xr + (g + 4 * c) * 8 + 4);
            }
            if (p.y1) {
                const bf16_t* yr = (const bf16_t*)p.y1 + (int64_t)(row0 + r) * D;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bf16x8 y = __builtin_bit_cast(bf16x8, *(const u32x4*)(yr + (g + 4 * c) * 8));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[c][0][e] += (float)y[e];
                        v[c][1][e] += (float)y[4 + e];
                    }
                }
            }
            ln_rows<NCH>(v, gam, bet, D, p.ln_eps, g, af[mf]);
        }
        MSTAMP(2);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == 0) p.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memtime();

        f32x4 acc2[2][NF2];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) acc2[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define LGKM(n)                                             \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0)
// before slab j of chunk c: if it opens a ring unit, wait until that unit has landed (at most the NUS-2
// later units in flight; 4*U DMA instructions per unit per wave), barrier, refill the slot freed by the
// previous unit (with the matching unit of the next pass once this pass is fully issued)
#define SLAB_SYNC(c, j)                                                                          \
    if ((DBG & 1) == 0 && (j) % U == 0) {                                                        \
        const int u = ((c) * SPC + (j)) / U;                                                     \
        const int later = cont ? NUS - 2 : (u + NUS - 2 < nunit ? u + NUS - 2 : nunit - 1) - u;  \
        wait_units<U>(later);                                                                    \
        __builtin_amdgcn_s_barrier();                                                            \
        if (u + NUS - 1 < nunit) issue_unit(u + NUS - 1);                                        \
        else if (cont) issue_unit(u + NUS - 1 - nunit);                                          \
    }
#define SLAB_ADDR(c, j) ((((((c) * SPC + (j)) / U) % NUS) * U + (j) % U) * SLAB_BYTES)

        for (int c = 0; c < nchunk; ++c) {
            // ================= fc1 chunk: acc1[128 rows, 128 hidden] =================
            if (c == 1) MSTAMP(5);
            f32x4 acc1[2][8];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int nf = 0; nf < 8; ++nf) acc1[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KS; ++kt) {
                SLAB_SYNC(c, kt);
                if ((DBG & 4) == 0 && my_frags > 0) {
                    const uint32_t a0 = lbase + SLAB_ADDR(c, kt) + foff[0];
                    const uint32_t a1 = lbase + SLAB_ADDR(c, kt) + foff[1];
                    // 4 groups per slab: 4 W fragments (ks, 4 column fragments) -> 8 MFMAs; two register sets
                    u32x4 wa[4], wb[4];
#define RD4(w, addr, base)                 \
    DSR128(w[0], addr, base + 0);          \
    DSR128(w[1], addr, base + 2048);       \
    DSR128(w[2], addr, base + 4096);       \
    DSR128(w[3], addr, base + 6144)
#define MM1(w, ks, q0)                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                        \
        Tr<bf16_t>::mma16(acc1[0][(q0) + j], w[j], af[0][kt * 2 + (ks)]);                  \
        Tr<bf16_t>::mma16(acc1[1][(q0) + j], w[j], af[1][kt * 2 + (ks)]);                  \
    }                                                                                      \
    __builtin_amdgcn_sched_barrier(0)
                    RD4(wa, a0, 0);
                    RD4(wb, a0, 8192);
                    LGKM(4); MM1(wa, 0, 0);
                    RD4(wa, a1, 0);
                    LGKM(4); MM1(wb, 0, 4);
                    RD4(wb, a1, 8192);
                    LGKM(4); MM1(wa, 1, 0);
                    LGKM(0); MM1(wb, 1, 4);
#undef RD4
#undef MM1
                }
            }
            if (c == 1) MSTAMP(6);
            // ================= bias + GELU, re-pack as fc2 operand fragments =================
            // acc1[mf][nf][e] = h[row li][hidden 128c + 16nf + 4g + e]; fragment f takes nf = 2f (slots 0-3) and 2f+1 (4-7)
            u32x4 hf[2][4];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    u32x4 o;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x4 u4 = acc1[mf][2 * f + h] + *(const f32x4*)(b1s + c * 128 + (2 * f + h) * 16 + 4 * g);
                        f32x2 lo = {u4[0], u4[1]}, hi = {u4[2], u4[3]};
                        if ((DBG & 2) == 0) {
                            lo = gelu2(lo);
                            hi = gelu2(hi);
                        }
                        o[2 * h] = pack_bf16x2(lo[0], lo[1]);
                        o[2 * h + 1] = pack_bf16x2(hi[0], hi[1]);
                    }
                    hf[mf][f] = o;
                }
            if (c == 1) MSTAMP(7);
            // ================= fc2 chunk: acc2 += h_c @ W2[:, chunk]^T =================
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    SLAB_SYNC(c, KS + kh * NG + ng);
                    if ((DBG & 4) == 0 && my_frags > 0) {
                        // 4 groups per slab: (fl, 4 output-column fragments) -> 8 x ds_read_b64, 8 MFMAs; two register sets
                        const uint32_t sb = lbase + SLAB_ADDR(c, KS + kh * NG + ng);
                        const uint32_t b00 = sb + f2off[0][0], b01 = sb + f2off[0][1], b10 = sb + f2off[1][0], b11 = sb + f2off[1][1];
                        u32x2 la[4], ha[4], lb[4], hb[4];
#define DSR64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define RD8(l, h, alo, ahi, base)                                       \
    DSR64(l[0], alo, base + 0);    DSR64(h[0], ahi, base + 0);          \
    DSR64(l[1], alo, base + 2048); DSR64(h[1], ahi, base + 2048);       \
    DSR64(l[2], alo, base + 4096); DSR64(h[2], ahi, base + 4096);       \
    DSR64(l[3], alo, base + 6144); DSR64(h[3], ahi, base + 6144)
#define MM2(l, h, fl, q0)                                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
        if (ng * 8 + (q0) + j < NF2) {                                                        \
            u32x4 wf;                                                                         \
            wf[0] = l[j][0]; wf[1] = l[j][1]; wf[2] = h[j][0]; wf[3] = h[j][1];               \
            Tr<bf16_t>::mma16(acc2[0][ng * 8 + (q0) + j], wf, hf[0][2 * kh + (fl)]);          \
            Tr<bf16_t>::mma16(acc2[1][ng * 8 + (q0) + j], wf, hf[1][2 * kh + (fl)]);          \
        }                                                                                     \
    }                                                                                         \
    __builtin_amdgcn_sched_barrier(0)
                        // (an asm read whose result is never used must not be issued: its destination would be
                        //  re-used while the data is still landing -- D = 192 has no fragments 12..15)
                        if (ng * 8 + 4 < NF2) {
                            RD8(la, ha, b00, b01, 0);
                            RD8(lb, hb, b00, b01, 8192);
                            LGKM(8); MM2(la, ha, 0, 0);
                            RD8(la, ha, b10, b11, 0);
                            LGKM(8); MM2(lb, hb, 0, 4);
                            RD8(lb, hb, b10, b11, 8192);
                            LGKM(8); MM2(la, ha, 1, 0);
                            LGKM(0); MM2(lb, hb, 1, 4);
                        } else {
                            RD8(la, ha, b00, b01, 0);
                            RD8(lb, hb, b10, b11, 0);
                            LGKM(8); MM2(la, ha, 0, 0);
                            LGKM(0); MM2(lb, hb, 1, 0);
                        }
#undef DSR64
#undef RD8
#undef MM2
                    }
                }
        }
        MSTAMP(3);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == 0) p.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime();
#undef DSR128
#undef LGKM
#undef SLAB_SYNC
#undef SLAB_ADDR

        // ---- epilogue: x <- x + y1 + acc2 + b2 (this workgroup owns its rows: in place, no other reader).
        //      All of a fragment's row loads are issued before the first use (the operand registers are free now).
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            const int r = (wave * 2 + mf) * 16 + li;
            if (r < nrows) {
                float* xr = p.x + (int64_t)(row0 + r) * D;
                const bf16_t* yr = p.y1 ? (const bf16_t*)p.y1 + (int64_t)(row0 + r) * D : nullptr;
                f32x4 xv[NF2];
                u32x2 yv[NF2];
#pragma unroll
                for (int nf = 0; nf < NF2; ++nf) xv[nf] = *(const f32x4*)(xr + nf * 16 + 4 * g);
                if (yr) {
#pragma unroll
                    for (int nf = 0; nf < NF2; ++nf) yv[nf] = *(const u32x2*)(yr + nf * 16 + 4 * g);
                }
#pragma unroll
                for (int nf = 0; nf < NF2; ++nf) {
                    const int n = nf * 16 + 4 * g;
                    f32x4 v = acc2[mf][nf] + *(const f32x4*)(b2s + n) + xv[nf];
                    if (yr) {
                        const bf16x4 y = __builtin_bit_cast(bf16x4, yv[nf]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)y[e];
                    }
                    *(f32x4*)(xr + n) = v;
                }
            }
        }
        MSTAMP(4);
        if (!cont) __syncthreads();  // the ring is re-primed from unit 0: every wave must be done reading it
        if (DBG & 1) __syncthreads();  // (no ring barriers in this debug build)
        tile = __builtin_amdgcn_readfirstlane(tile_s[(seq + 1) & 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: units of a pass that never runs)
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
}

template <int KS, int DBG = 0>
int launch(const MlpParams& p_in, hipStream_t st) {
    MlpParams p = p_in;
    constexpr int D = KS * 64;
    const int lds = NSLOT * SLAB_BYTES + (3 * D + p.hidden) * 4 + 16;
    auto k = mlp_kernel<KS, DBG>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(mlp) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("mlp: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    const int ncu = once.ncu[dev];
    // whole rounds of #CU workgroups take 128 rows each; a last partial round that would be less than an
    // eighth full is cut into 16-row tiles (one active wave each: such a tile costs about half a full one)
    const int tiles = (p.M + TMR - 1) / TMR;
    const int rem = tiles % ncu;
    // (a launch of at most four tiles -- the second-level ViT on one region, the [CLS] rows of the pruned block -- is all 16-row tiles:
    //  the same weight pass per workgroup, a fraction of the row work, 8x the CUs of an otherwise idle GPU)
    const int tail_tiles = tiles <= 4 ? tiles : ((tiles > ncu && rem > 0 && rem <= ncu / 8) ? rem : 0);
    p.full_tiles = tiles - tail_tiles;
    const int tail_rows = p.M - p.full_tiles * TMR;
    p.ntiles = p.full_tiles + (tail_rows > 0 ? (tail_rows + 15) / 16 : 0);
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    // long launches (>= 6 tiles per workgroup) start their 4 workgroup groups a quarter of a tile time apart
    const int stag_us = 20;
    p.stagger = (p.full_tiles >= 6 * ncu) ? stag_us * 100 : 0;
    if (hipMemsetAsync(p.counter, 0, sizeof(int), st) != hipSuccess) {
        hipt_set_error("mlp: hipMemsetAsync(counter) failed");
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
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, p);
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
        double ph[4] = {0, 0, 0, 0};
        const int nb = grid < 256 ? grid : 256;
        for (int b = 0; b < nb; ++b)
            for (int k2 = 0; k2 < 4; ++k2) ph[k2] += (double)(h[b * 16 + k2 + 1] - h[b * 16 + k2]) * 0.01 / nb;
        double c1 = 0, c2 = 0, ghz = 0;
        for (int b = 0; b < nb; ++b) {
            c1 += (double)(h[b * 16 + 6] - h[b * 16 + 5]) * 0.01 / nb;
            c2 += (double)(h[b * 16 + 7] - h[b * 16 + 6]) * 0.01 / nb;
            ghz += (double)(h[b * 16 + 9] - h[b * 16 + 8]) / (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.1 / nb;
        }
        fprintf(stderr, "[mlp KS=%d dbg=%d hidden=%d grid=%d full=%d stagger=%d] total %.1f us | first tiles: stage %.1f, Aload+LN %.1f, chunks %.1f (chunk1: fc1 %.2f, gelu %.2f; %.2f GHz), epilogue %.1f\n", KS,
                DBG, p.hidden, grid, p.full_tiles, p.stagger, (double)(t4 - t0) * 0.01, ph[0], ph[1], ph[2], c1, c2, ghz, ph[3]);
    }
#endif
    return HIPT_OK;
}

}  // namespace

bool hipt_mlp_supported(int dtype, int D, int hidden) {
    return dtype == HIPT_BF16 && (D == 384 || D == 192) && hidden % 128 == 0 && hidden <= 4096;
}

int hipt_mlp_launch(const MlpParams& p, hipStream_t st) {
    HIPT_CHECK_ARG(p.M > 0 && p.x && p.w1 && p.w2 && p.b1 && p.b2 && p.ln_w && p.ln_b && p.counter, "mlp: null/empty argument");
    HIPT_CHECK_ARG(((uintptr_t)p.x % 16) == 0 && ((uintptr_t)p.w1 % 16) == 0 && ((uintptr_t)p.w2 % 16) == 0 &&
                       ((uintptr_t)p.y1 % 16) == 0,
                   "mlp: 16-byte alignment required");
    // the streaming kernel (mlp16.hip) runs from its packed weight image: callers without one get the generic kernel below
    if (!hipt_generic_only() && p.wpk && (p.wpk_fmt == 2 || p.wpk_fmt == 3) && hipt_mlp16_supported(HIPT_BF16, p.D, p.hidden)) return hipt_mlp16_launch(p, st);
    HIPT_CHECK_ARG(p.img == 0 && !p.xn_out && !p.fold, "mlp: activation images / the chained LayerNorm exist only in the streaming kernel (img=%d)", p.img);
    if (p.D == 384) return launch<6>(p, st);
    if (p.D == 192) return launch<3>(p, st);
    hipt_set_error("mlp: D=%d not in {192, 384}", p.D);
    return HIPT_E_UNSUPPORTED;
}
