// FUSED MLP SUB-BLOCK for D = 384 (ViT-256), fc2 COLUMN-OWNED:   x <- x + y1 + fc2( GELU( fc1( LN2(x + y1) ) ) )
//   (Block.forward second half, HIPT_4K/vision_transformer.py:151 with Mlp.forward :98-104.)
//
// mlp32.hip with one change of ownership.  There every wave owns 32 rows through both Linears, so each of the four waves reads
// EVERY weight fragment from the LDS ring: 1 KiB of LDS per MFMA, and both matrices go L2 -> LDS -> registers.  The ablations of
// round 2 say that this traffic, not the issue slots it takes, is what the kernel pays for: the chip is
// power-limited under it (1.6 GHz in the chunk phases; 2.2-2.3 GHz with half the fragment reads or without the weight DMA).  Here
//   * fc1 stays row-owned (the LayerNorm'd rows of a wave are its B operand in registers; W1 units stream through a two-slot ring);
//   * fc2 is COLUMN-owned: wave w accumulates output tiles 3 w .. 3 w + 2 for all 128 rows (the same 192 accumulator registers).
//     Its A operand -- W2 fragments 12 w .. 12 w + 11 of a unit, which are exactly the 12 KiB the wave used to DMA -- is loaded
//     from L2 straight into registers and used for FOUR MFMAs (one per 32-row block); its B operand, the GELU'd hidden tile, is
//     exchanged through LDS: every wave writes the operand fragments of its 32 rows (the accumulator-as-operand registers of
//     mlp32.hip, 4 KiB per half-chunk) and reads those of all four row blocks, each fragment for THREE MFMAs.
//   Per tile and wave: 1152 + 384 fragment reads instead of 2304, 24 ring units instead of 48, W2 never touches LDS.
// Same packed weight image as mlp32.hip (format 1), same activation images; image forms only (img == 3: every block of a forward
// but the first, which stays with mlp32.hip).
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "mlp_common.h"
#include "pipe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int D = 384, NCH = 12, NKS = 24, TMR = 128;
constexpr int UNIT = 48 * 1024;    // a weight unit of the image = 48 fragments of 1 KiB
constexpr int HSLOT = 16 * 1024;   // the hidden exchange of one half-chunk: 4 row blocks x 4 k-steps x 1 KiB

__device__ __forceinline__ void mma32(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

#ifndef PSTAMP_SEQ
#define PSTAMP_SEQ 0
#endif
#define PSTAMP(k)                                                                                                    \
    do {                                                                                                             \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

enum { KA = 0, KB = 1 };

template <int DUMMY = 0>
__global__ __launch_bounds__(256, 1) void mlp_co_kernel(const MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* hb = smem + 2 * UNIT;                         // [2][HSLOT] hidden exchange, slot = half of the chunk
    float* gam = (float*)(smem + 2 * UNIT + 2 * HSLOT);
    float* bet = gam + D;
    float* b2s = bet + D;
    float* b1s = b2s + D;                  // [hidden]
    int* tile_s = (int*)(b1s + p.hidden);  // [2] tile handed to this workgroup, double-buffered by parity
    float* gam1 = (float*)(tile_s + 4);    // next block's LayerNorm-1 (gamma | beta), if p.xn_out
    float* lnx = gam1 + 2 * D;             // [2][4][128] row statistics exchanged between the waves in the epilogue

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;        // the row phase's 16-row fragment view: lane (li, g) owns chunks g + 4c
    const int h = lane >> 5, m = (lane >> 4) & 1;   // the MFMA view: lane = 32 h + 16 m + li holds row (fragment m, li), k half h
    const int nchunk = p.hidden / 128;
    const int upt = 4 * nchunk;                     // units of the image = phases of a tile pass

    // ---- W1 units through the ring: wave w issues pieces 12 w .. 12 w + 11 of a unit, six per phase (four consecutive pieces share
    // one M0, the piece is selected by the instruction's immediate offset: see mlp32.hip)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 2 * p.hidden * D * 2, 0x00020000);
    const uint32_t ilane = (uint32_t)(12 * wave * 1024 + lane * 16);
    // The stream: A unit k of a pass (k = 0 .. 2 n - 1, image position k ? 2 k - 1 : 0, ring slot k & 1) is read by the pass's A phase k.
    // Phase j of a pass (A0(0) A1(0) B0(0) | A0(c) B1(c-1) A1(c) B0(c) ... | B1(n-1)) issues six pieces: the first six of unit
    // (j + 4) / 2 when j is even, the second six of unit (j + 3) / 2 when odd -- a unit goes out right behind the last read of
    // its slot's previous tenant and has one whole phase to land; units past 2 n - 1 are the next pass's (the stream does not stop
    // between tiles).  The one exception is unit 2: its slot is read by phase 0, so phase 0 issues nothing and phase 1 all twelve.
    // Always into the slot the current / last A phase does NOT read.
    int aslot = 0;  // ring slot of the A unit the current (or, in a B phase, the next) A phase reads
    auto dma_piece = [&](auto T_, int pos, int slot) __attribute__((always_inline)) {
        constexpr int t = decltype(T_)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + slot * UNIT + (12 * wave + (t & ~3)) * 1024), 16, ilane, pos * UNIT + (t & ~3) * 1024,
                                                 (t & 3) * 1024, 0);
    };
    enum { PC_NONE = 0, PC_FIRST = 1, PC_SECOND = 2, PC_ALL = 3 };
    // W2 fragment (k-step q, output tile 3 w + ot) of the B unit at image position pos: fragment 12 w + 4 ot + q
    auto w2_load = [&](int pos, int q, int ot) __attribute__((always_inline)) {
        return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ilane, pos * UNIT + (4 * ot + q) * 1024, 0));
    };

    for (int i = tid; i < D; i += 256) {
        gam[i] = p.ln_w[i];
        bet[i] = p.ln_b[i];
        b2s[i] = p.b2[i];
        if (p.xn_out) {
            gam1[i] = p.ln_next_w[i];
            gam1[D + i] = p.ln_next_b[i];
        }
    }
    for (int i = tid; i < p.hidden; i += 256) b1s[i] = p.b1[i];
    // (the tile queue resets itself: a launch makes grid + ntiles fetches, the one that draws the last number stores 0 -- nobody
    //  fetches after it -- so that a caller running a chain of these kernels zeroes the counter once, not once per launch)
    const int last_fetch = p.ntiles + (int)gridDim.x - 1;
    if (tid == 0) {
        const int t0 = atomicAdd(p.counter, 1);
        if (t0 == last_fetch) *p.counter = 0;
        tile_s[0] = t0;
    }
    __syncthreads();
    int tile = __builtin_amdgcn_readfirstlane(tile_s[0]);
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memrealtime();

    const uint32_t lbase = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    const uint32_t fbase = lbase + lane * 16;                                         // + slot * UNIT + fragment * 1024
    const uint32_t hbase = (uint32_t)(uintptr_t)(LDS_AS char*)hb + lane * 16;         // + half * HSLOT + (4 rb + q) * 1024
    const uint32_t b1base = (uint32_t)(uintptr_t)(LDS_AS char*)b1s + 16 * h;          // b1[Hb + 32 U + 8 q + 4 h ..]: + (Hb + 32 U + 8 q) * 4
    const uint32_t tsbase = (uint32_t)(uintptr_t)(LDS_AS char*)tile_s;
    const uint32_t gbase = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 32 * g;           // (row phase: 16-row fragment view)
    const uint32_t b2base = (uint32_t)(uintptr_t)(LDS_AS char*)b2s + 16 * h;          // b2[32 O + 8 q + 4 h ..]: + (32 O + 8 q) * 4
    const uint32_t g1base = (uint32_t)(uintptr_t)(LDS_AS char*)gam1 + 16 * h;         // next LN-1 gamma (beta: + D * 4)
    const uint32_t lxbase = (uint32_t)(uintptr_t)(LDS_AS char*)lnx + (lane & 31) * 4;  // + (set * 4 + wave) * 512 + rb * 128

    // ---- prime the ring: A units 0 and 1 of the first pass ----
    if (tile < p.ntiles) {
        sfor<0, 12>([&](auto T_) __attribute__((always_inline)) { dma_piece(T_, 0, 0); });
        sfor<0, 12>([&](auto T_) __attribute__((always_inline)) { dma_piece(T_, 1, 1); });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment registers: two sets of 4.  A phases: the four W1 fragments of a 4-MFMA group; B phases: the hidden fragments of
    // the four row blocks for one k-step
    u32x4 wA[2][4];
    auto rd4 = [&](auto SET_, uint32_t a, auto O0_, auto STRIDE_) __attribute__((always_inline)) {
        constexpr int set = decltype(SET_)::value, o0 = decltype(O0_)::value, st = decltype(STRIDE_)::value;
        const uint32_t a_ = a;
        u32x4 &d0 = wA[set][0], &d1 = wA[set][1], &d2 = wA[set][2], &d3 = wA[set][3];
        DSR128(d0, a_, o0);
        DSR128(d1, a_, o0 + st);
        DSR128(d2, a_, o0 + 2 * st);
        DSR128(d3, a_, o0 + 3 * st);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    typedef std::integral_constant<int, 1024> S1K;
    typedef std::integral_constant<int, 4096> S4K;

    f32x4 bq[2][4];  // fc1 bias an A phase starts from: tile U, quad q: b1[off + 32 U + 8 q + 4 h + e], read one phase ahead
    auto bias_rd = [&](int off) __attribute__((always_inline)) {  // 8 reads, no wait: covered by the next counted wait
        const uint32_t a = b1base + off * 4;
        f32x4 &q0 = bq[0][0], &q1 = bq[0][1], &q2 = bq[0][2], &q3 = bq[0][3], &q4 = bq[1][0], &q5 = bq[1][1], &q6 = bq[1][2], &q7 = bq[1][3];
        DSR128(q0, a, 0);
        DSR128(q1, a, 32);
        DSR128(q2, a, 64);
        DSR128(q3, a, 96);
        DSR128(q4, a, 128);
        DSR128(q5, a, 160);
        DSR128(q6, a, 192);
        DSR128(q7, a, 224);
    };

    for (int seq = 0; tile < p.ntiles; ++seq) {
        // tiles [0, full_tiles): 128 rows each; then 16-row tail tiles
        int row0, nrows;
        if (tile < p.full_tiles) {
            row0 = tile * TMR;
            nrows = TMR;
        } else {
            row0 = p.full_tiles * TMR + (tile - p.full_tiles) * 16;
            nrows = 16;
        }
        nrows = (p.M - row0) < nrows ? (p.M - row0) : nrows;
        PSTAMP(0);
        int nt_req = 0;
        if (tid == 0) nt_req = atomicAdd(p.counter, 1);  // next tile: handed to LDS behind the first row loads, read after a ring barrier
        int tile_next = 0;

        u32x4 X[NKS];  // the fc1 B operand: k-step s = 2 c + p, lane half h: columns 32 c + 16 p + 8 (j >> 2) + 4 h + (j & 3)
        f32x16 acc2[3][4];  // [ot][rb]: output tile 3 w + ot of row block rb: register 4 q + e = column 32 O + 8 q + 4 h + e of row 32 rb + (lane & 31)
#pragma unroll
        for (int o = 0; o < 3; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[o][r][e] = 0.f;
        f32x16 acc1[2][2];  // [half][tile U]: hidden (reg & 3) + 8 (reg >> 2) + 4 h of tile U for this lane's row
        u32x4 w2[4][3];     // [k-step q][ot]: the W2 fragments of the B unit the next B phase works on
        u32x4 hst;          // four GELU'd pairs on their way to the exchange: one 16-byte store per fragment and lane

        // one 2-element GELU: unit u (0..15) of half GH -> word (u & 3) of fragment u >> 2 (= 2 U + s') of this wave's row block
        auto gelu_unit = [&](auto GH_, auto U_) __attribute__((always_inline)) {
            constexpr int gh = decltype(GH_)::value, u = decltype(U_)::value;
            constexpr int tl = u >> 3, pi = u & 7;
            hst[u & 3] = pack_bf16x2(gelu1(acc1[gh][tl][2 * pi]), gelu1(acc1[gh][tl][2 * pi + 1]));
            if constexpr ((u & 3) == 3) {
                const uint32_t a = hbase + (uint32_t)wave * 4096;
                const u32x4 v = hst;
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(gh * HSLOT + (u >> 2) * 1024) : "memory");
            }
        };

        // ---- one phase = 12 groups of 4 MFMAs ----
        // KA, half H: fc1 of hidden half H of a chunk into acc1[H][.] (started from the bias in bq), W1 unit in ring slot aslot.
        // KB, half H: fc2 over the hidden half H (exchange slot H): k-step q = group / 3, output tile ot = group % 3, one MFMA per row block.
        // GH/GSEC: GELU units of half GH, first (0) or second (1) eight, one per group 3..10; GH = -1: none.
        // NX: what the last group reads ahead for the next phase: 0 nothing (the caller synchronises), 1 the next A phase's first
        //     fragments + bias at b1s offset nb, 2 the next B phase's first hidden fragments (half 1 - H).
        // RL (KB): reload w2 for the B unit at image position wn behind each k-step.
        // PC / ppos: which pieces of the A unit at image position ppos this phase issues (see the stream above).
        // VW: >= 0: before the barrier, wait until at most VW of this wave's vector-memory operations are in flight -- the phases
        //     with an even index: everything issued before this phase has landed then, i.e. the A unit of the phase after the next.
        auto phase = [&](auto KIND_, auto H_, auto GH_, auto GSEC_, auto NX_, int nb, auto RL_, int wn, auto PC_, int ppos, auto VW_) __attribute__((always_inline)) {
            constexpr int kind = decltype(KIND_)::value, hh = decltype(H_)::value, gh = decltype(GH_)::value;
            constexpr int gsec = decltype(GSEC_)::value, nx = decltype(NX_)::value, rl = decltype(RL_)::value;
            constexpr int pc = decltype(PC_)::value, vw = decltype(VW_)::value;
            const uint32_t sa = fbase + aslot * UNIT;
            sfor<0, 12>([&](auto G_) __attribute__((always_inline)) {
                constexpr int gg = decltype(G_)::value;
                constexpr int q = gg / 3, ot = gg % 3;
                constexpr int set = kind == KA ? (gg & 1) : (q & 1);
                typedef std::integral_constant<int, set ^ 1> NS;
                // (1) reads ahead
                if constexpr (gg == 11) {
                    LGKM(0);  // this group's fragments; and my exchange stores are in LDS before the barrier
                    if constexpr (vw >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(vw) : "memory");
                    __builtin_amdgcn_s_barrier();
                    if constexpr (kind == KA) {
                        aslot ^= 1;
                    }
                    if constexpr (nx == 1) {
                        rd4(I0{}, fbase + aslot * UNIT, I0{}, S1K{});
                        bias_rd(nb);
                    } else if constexpr (nx == 2) {
                        rd4(I0{}, hbase, std::integral_constant<int, (1 - hh) * HSLOT>{}, S4K{});  // (an A phase of half H is followed by the B phase of the other half)
                    }
                } else if constexpr (kind == KA) {
                    rd4(NS{}, sa, std::integral_constant<int, (4 * (gg + 1)) * 1024>{}, S1K{});
                    LGKM(4);
                } else if constexpr (ot == 0) {
                    if constexpr (q < 3) {
                        rd4(NS{}, hbase, std::integral_constant<int, hh * HSLOT + (q + 1) * 1024>{}, S4K{});
                        LGKM(4);
                    } else {
                        LGKM(0);
                    }
                }
                // (2) 4 MFMAs
                if constexpr (kind == KA) {
                    if constexpr (gg == 0) {
                        f32x4 &q0 = bq[0][0], &q1 = bq[0][1], &q2 = bq[0][2], &q3 = bq[0][3], &q4 = bq[1][0], &q5 = bq[1][1], &q6 = bq[1][2], &q7 = bq[1][3];
                        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
                        f32x16 t0, t1;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                t0[4 * qq + e] = bq[0][qq][e];
                                t1[4 * qq + e] = bq[1][qq][e];
                            }
                        mma32(t0, wA[set][0], X[0]);
                        mma32(t1, wA[set][1], X[0]);
                        mma32(t0, wA[set][2], X[1]);
                        mma32(t1, wA[set][3], X[1]);
                        acc1[hh][0] = t0;
                        acc1[hh][1] = t1;
                    } else {
                        mma32(acc1[hh][0], wA[set][0], X[2 * gg]);
                        mma32(acc1[hh][1], wA[set][1], X[2 * gg]);
                        mma32(acc1[hh][0], wA[set][2], X[2 * gg + 1]);
                        mma32(acc1[hh][1], wA[set][3], X[2 * gg + 1]);
                    }
                } else {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) mma32(acc2[ot][rb], w2[q][ot], wA[set][rb]);
                }
                // (3) the weight stream: this phase's pieces of the A units ahead, two per group; B phases reload their W2 window
                if constexpr (pc == PC_ALL && gg < 6) {
                    dma_piece(std::integral_constant<int, 2 * gg>{}, ppos, aslot ^ 1);
                    dma_piece(std::integral_constant<int, 1 + 2 * gg>{}, ppos, aslot ^ 1);
                } else if constexpr ((pc == PC_FIRST || pc == PC_SECOND) && gg < 3) {
                    dma_piece(std::integral_constant<int, (pc == PC_SECOND ? 6 : 0) + 2 * gg>{}, ppos, aslot ^ 1);
                    dma_piece(std::integral_constant<int, (pc == PC_SECOND ? 7 : 1) + 2 * gg>{}, ppos, aslot ^ 1);
                }
                if constexpr (kind == KB && rl != 0 && ot == 2) {
#pragma unroll
                    for (int o = 0; o < 3; ++o) w2[q][o] = w2_load(wn, q, o);
                }
                if constexpr (gh >= 0 && gg >= 3 && gg <= 10) {
                    gelu_unit(std::integral_constant<int, (gh >= 0 ? gh : 0)>{}, std::integral_constant<int, 8 * gsec + gg - 3>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        typedef std::integral_constant<int, -1> IM1;
        typedef std::integral_constant<int, 2> I2;
        typedef std::integral_constant<int, KA> TA;
        typedef std::integral_constant<int, KB> TB;

        // ---- row phase: v = x + y1 -> LN2 -> operand fragments (activation images; as mlp32.hip) ----
        {
            u32x4 af[2][NCH];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                // whole fragments only (M % 16 == 0); a fragment past the tile's end re-reads fragment 0 of the tile (never stored)
                const int fr = (wave * 2 + mf) * 16 < nrows ? (wave * 2 + mf) * 16 : 0;
                const float* xr = p.x + (int64_t)(row0 + fr) * D + lane * 4;
                f32x4 v[NCH][2];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    v[c][0] = *(const f32x4*)(xr + c * 512);
                    v[c][1] = *(const f32x4*)(xr + c * 512 + 256);
                }
                if (p.y1) {
                    const bf16_t* yr = (const bf16_t*)p.y1 + (int64_t)(row0 + fr) * D + lane * 8;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const bf16x8 y = __builtin_bit_cast(bf16x8, *(const u32x4*)(yr + c * 512));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[c][0][e] += (float)y[e];
                            v[c][1][e] += (float)y[4 + e];
                        }
                    }
                }
                if (mf == 0 && tid == 0) {
                    asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt_req) : "memory");
                    if (nt_req == last_fetch) *p.counter = 0;
                }
                ln_rows_lds<D, NCH>(v, gbase, p.ln_eps, af[mf]);
                if (mf == 0) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        u32x4& a2 = af[0][c];
                        asm volatile("" : "+a"(a2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // 16-row fragments -> the 32-row B operand in accumulator column order (mlp32.hip, to_operand)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                u32x4 e4, o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(af[0][c][e], af[1][c][e], false, false);
                    e4[e] = sw[0];
                    o4[e] = sw[1];
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const auto s0 = __builtin_amdgcn_permlane32_swap(e4[e], e4[2 + e], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(o4[e], o4[2 + e], false, false);
                    e4[e] = s0[0];
                    e4[2 + e] = s0[1];
                    o4[e] = s1[0];
                    o4[2 + e] = s1[1];
                }
                X[2 * c] = u32x4{e4[0], e4[1], o4[0], o4[1]};
                X[2 * c + 1] = u32x4{e4[2], e4[3], o4[2], o4[3]};
            }
        }
        PSTAMP(2);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memtime();

        // the W2 window of the pass's first B unit (image position 2): two A phases to land
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int o = 0; o < 3; ++o) w2[q][o] = w2_load(2, q, o);
        // first fragments and bias of the pass
        rd4(I0{}, fbase + aslot * UNIT, I0{}, S1K{});
        bias_rd(0);
        // chunk 0 (peeled).  Its half-0 GELUs have only A1(0) to hide in: the second eight run bare, behind a barrier of their own
        typedef std::integral_constant<int, PC_NONE> P0;
        typedef std::integral_constant<int, PC_FIRST> P1;
        typedef std::integral_constant<int, PC_SECOND> P2;
        typedef std::integral_constant<int, PC_ALL> P3;
        typedef std::integral_constant<int, 12> V12;  // (phase 0: the twelve W2 loads above may still be in flight)
        typedef std::integral_constant<int, 18> V18;  // (even B phases: their own six pieces and twelve W2 loads)
        phase(TA{}, I0{}, IM1{}, I0{}, I1{}, 64, I0{}, 0, P0{}, 0, V12{});   // A0(0)
        phase(TA{}, I1{}, I0{}, I0{}, I0{}, 0, I0{}, 0, P3{}, 3, IM1{});      // A1(0) + first eight GELUs of half 0; all of A unit 2
        sfor<8, 16>([&](auto U_) __attribute__((always_inline)) { gelu_unit(I0{}, U_); });
        __builtin_amdgcn_sched_barrier(0);
        {
            // the tile after this one (handed over before the first ring barrier); the wait also covers the exchange stores above
            int nt;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nt) : "v"(tsbase + 4 * ((seq + 1) & 1)) : "memory");
            tile_next = __builtin_amdgcn_readfirstlane(nt);
            __builtin_amdgcn_s_barrier();
            rd4(I0{}, hbase, I0{}, S4K{});
        }
        phase(TB{}, I0{}, I1{}, I0{}, I1{}, 128, I1{}, 4, P1{}, 5, V18{});   // B0(0) + first eight of half 1; reloads B1(0)
        for (int c = 1; c < nchunk - 1; ++c) {
            phase(TA{}, I0{}, I1{}, I1{}, I2{}, 0, I0{}, 0, P2{}, 4 * c + 1, IM1{});                       // A0(c) + second eight of half 1 of chunk c-1
            phase(TB{}, I1{}, I0{}, I0{}, I1{}, c * 128 + 64, I1{}, 4 * c + 2, P1{}, 4 * c + 3, V18{});    // B1(c-1) + first eight of half 0 of chunk c
            phase(TA{}, I1{}, I0{}, I1{}, I2{}, 0, I0{}, 0, P2{}, 4 * c + 3, IM1{});                       // A1(c) + second eight of half 0
            phase(TB{}, I0{}, I1{}, I0{}, I1{}, (c + 1) * 128, I1{}, 4 * c + 4, P1{}, 4 * c + 5, V18{});   // B0(c) + first eight of half 1
        }
        {   // the last chunk (peeled: no branch inside the loop): its pieces are the next pass's A units 0 and 1, B1(n-1) ends the image
            const int c = nchunk - 1;
            phase(TA{}, I0{}, I1{}, I1{}, I2{}, 0, I0{}, 0, P2{}, 4 * c + 1, IM1{});
            phase(TB{}, I1{}, I0{}, I0{}, I1{}, c * 128 + 64, I1{}, 4 * c + 2, P1{}, 0, V18{});
            phase(TA{}, I1{}, I0{}, I1{}, I2{}, 0, I0{}, 0, P2{}, 0, IM1{});
            phase(TB{}, I0{}, I1{}, I0{}, I0{}, 0, I1{}, upt - 1, P1{}, 1, V18{});
        }
        // tail: second eight of the last half 1 bare, their barrier, then B1(last)
        sfor<8, 16>([&](auto U_) __attribute__((always_inline)) { gelu_unit(I1{}, U_); });
        __builtin_amdgcn_sched_barrier(0);
        LGKM(0);
        __builtin_amdgcn_s_barrier();
        rd4(I0{}, hbase, std::integral_constant<int, HSLOT>{}, S4K{});
        phase(TB{}, I1{}, IM1{}, I0{}, I0{}, 0, I0{}, 0, P2{}, 1, IM1{});
        PSTAMP(3);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime();

        // ---- epilogue: x <- x + y1 + acc2 + b2 on this wave's 96 columns of all 128 rows (images: fragment 2 rb + m of the tile,
        //      fp32 piece (O, q): fb + 512 O + 256 h + 64 q + 4 li; bf16 piece: fb + 512 O + 128 q + 8 li + 4 h)
        {
#pragma clang fp contract(off)
            float rs_[4];
            f32x4 xv[6][4];
            u32x2 yv[6][4];
            auto ld_batch = [&](auto B_) __attribute__((always_inline)) {
                constexpr int rb = decltype(B_)::value, s0 = (rb & 1) * 3;
                const int fr = rb * 32 + m * 16;
                const int64_t fb = (int64_t)(row0 + (fr < nrows ? fr : 0)) * D;
                const float* xl = p.x + fb + 256 * h + 4 * li + 512 * 3 * wave;
                const bf16_t* yr = (const bf16_t*)p.y1 + fb + 8 * li + 4 * h + 512 * 3 * wave;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        xv[s0 + i][q] = *(const f32x4*)(xl + 512 * i + 64 * q);
                        yv[s0 + i][q] = p.y1 ? *(const u32x2*)(yr + 512 * i + 128 * q) : u32x2{0u, 0u};
                    }
            };
            ld_batch(I0{});
            sfor<0, 4>([&](auto B_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int rb = decltype(B_)::value, s0 = (rb & 1) * 3;
                if constexpr (rb < 3) ld_batch(std::integral_constant<int, rb + 1>{});
                const int fr = rb * 32 + m * 16;
                const bool live = fr < nrows;
                float* xs = p.x + (int64_t)(row0 + (live ? fr : 0)) * D + 256 * h + 4 * li + 512 * 3 * wave;
                float rs = 0.f;
                sfor<0, 3>([&](auto I_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int i = decltype(I_)::value;
                    f32x4 bb[4];
                    const uint32_t ba = b2base + (uint32_t)wave * (3 * 128);
                    f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                    DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, i * 128, i * 128 + 32, i * 128 + 64, i * 128 + 96);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bf16x4 y = __builtin_bit_cast(bf16x4, yv[s0 + i][q]);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ((acc2[i][rb][4 * q + e] + bb[q][e]) + xv[s0 + i][q][e]) + (float)y[e];
                        if (live) *(f32x4*)(xs + 512 * i + 64 * q) = v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc2[i][rb][4 * q + e] = v[e];
                            rs += v[e];
                        }
                    }
                });
                rs_[rb] = rs;
            });
            if (p.xn_out) {
                // LayerNorm-1 of the next block on the finished rows.  A row's 384 columns sit in four waves (and two lane halves):
                // partial sums go through LDS, in wave order, so that a row's result does not depend on where it sits
                float mean[4], rstd[4];
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    rs_[rb] += __shfl_xor(rs_[rb], 32, 64);
                    if (h == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lxbase + (uint32_t)wave * 512), "v"(rs_[rb]), "n"(rb * 128) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    float s0, s1, s2, s3;
                    asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                                 : "v"(lxbase), "n"(rb * 128), "n"(512 + rb * 128), "n"(1024 + rb * 128), "n"(1536 + rb * 128)
                                 : "memory");
                    mean[rb] = (((s0 + s1) + s2) + s3) * (1.0f / D);
                    float qs = 0.f;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const float a = acc2[i][rb][e] - mean[rb];
                            qs = __builtin_fmaf(a, a, qs);
                        }
                    qs += __shfl_xor(qs, 32, 64);
                    if (h == 0) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(lxbase + (uint32_t)wave * 512), "v"(qs), "n"(2048 + rb * 128) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    float s0, s1, s2, s3;
                    asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3)
                                 : "v"(lxbase), "n"(2048 + rb * 128), "n"(2048 + 512 + rb * 128), "n"(2048 + 1024 + rb * 128), "n"(2048 + 1536 + rb * 128)
                                 : "memory");
                    rstd[rb] = 1.0f / sqrtf((((s0 + s1) + s2) + s3) * (1.0f / D) + p.ln_eps);
                }
                sfor<0, 3>([&](auto I_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int i = decltype(I_)::value;
                    f32x4 gq[4], bqv[4];
                    const uint32_t ga = g1base + (uint32_t)wave * (3 * 128);
                    f32x4 &g0 = gq[0], &g1 = gq[1], &g2 = gq[2], &g3 = gq[3], &b0 = bqv[0], &b1_ = bqv[1], &b2_ = bqv[2], &b3 = bqv[3];
                    DSR128X4_WAIT(g0, g1, g2, g3, ga, i * 128, i * 128 + 32, i * 128 + 64, i * 128 + 96);
                    DSR128X4_WAIT(b0, b1_, b2_, b3, ga, D * 4 + i * 128, D * 4 + i * 128 + 32, D * 4 + i * 128 + 64, D * 4 + i * 128 + 96);
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        const int fr = rb * 32 + m * 16;
                        const bool live = fr < nrows;
                        bf16_t* nr = (bf16_t*)p.xn_out + (int64_t)(row0 + (live ? fr : 0)) * D + 8 * li + 4 * h + 512 * 3 * wave;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float y[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc2[i][rb][4 * q + e] - mean[rb]) * rstd[rb], gq[q][e], bqv[q][e]);
                            u32x2 o2;
                            o2[0] = pack_bf16x2(y[0], y[1]);
                            o2[1] = pack_bf16x2(y[2], y[3]);
                            if (live) *(u32x2*)(nr + 512 * i + 128 * q) = o2;
                        }
                    }
                });
            }
        }
        PSTAMP(4);
        tile = tile_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: pieces of a pass that never runs)
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

bool hipt_mlp_co_supported(int dtype, int D_, int hidden) {
    return dtype == HIPT_BF16 && D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536;
}

int hipt_mlp_co_launch(const MlpParams& p_in, hipStream_t st) {
    MlpParams p = p_in;
    const int lds = 2 * UNIT + 2 * HSLOT + (3 * D + p.hidden) * 4 + 16 + 2 * D * 4 + 2 * 4 * 128 * 4;
    if (!p.wpk || p.wpk_fmt != 1 || p.img != 3 || p.M % 16 != 0 || p.fold) {
        hipt_set_error("mlp_co: needs the packed weights of format 1 and activation images on both sides (img=%d, M=%d, fold=%d)", p.img, p.M, p.fold);
        return HIPT_E_BADARG;
    }
    auto k = mlp_co_kernel<0>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(mlp_co) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("mlp_co: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    const int ncu = once.ncu[dev];
    // (tiling as in mlp32.hip: whole 128-row tiles; a small leftover round of a short launch is cut into 16-row tiles)
    const int tiles = (p.M + TMR - 1) / TMR;
    const int rem = tiles % ncu;
    const int tail_tiles = (tiles > ncu && tiles <= 4 * ncu + ncu / 8 && rem > 0 && rem <= ncu / 8) ? rem : 0;
    p.full_tiles = tiles - tail_tiles;
    const int tail_rows = p.M - p.full_tiles * TMR;
    p.ntiles = p.full_tiles + (tail_rows > 0 ? (tail_rows + 15) / 16 : 0);
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    p.stagger = 0;
    if (!p.counter_zeroed && hipMemsetAsync(p.counter, 0, sizeof(int), st) != hipSuccess) {
        hipt_set_error("mlp_co: hipMemsetAsync(counter) failed");
        return HIPT_E_LAUNCH;
    }
#ifdef HIPT_DEBUG_STAMPS
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
#ifdef HIPT_DEBUG_STAMPS
    if (want_stamps && grid <= 4096) {
        static unsigned long long hbuf[4096 * 16];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(hbuf, dbuf, (size_t)grid * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t4 = 0;
        for (int b = 0; b < grid; ++b) {
            if (hbuf[b * 16 + 11] < t0) t0 = hbuf[b * 16 + 11];
            if (hbuf[b * 16 + 10] > t4) t4 = hbuf[b * 16 + 10];
        }
        double pro = 0, chunks = 0, epi = 0, ghz = 0;
        for (int b = 0; b < grid; ++b) {
            pro += (double)(hbuf[b * 16 + 2] - hbuf[b * 16 + 0]) * 0.01 / grid;
            chunks += (double)(hbuf[b * 16 + 3] - hbuf[b * 16 + 2]) * 0.01 / grid;
            epi += (double)(hbuf[b * 16 + 4] - hbuf[b * 16 + 3]) * 0.01 / grid;
            ghz += (double)(hbuf[b * 16 + 9] - hbuf[b * 16 + 8]) / (double)(hbuf[b * 16 + 3] - hbuf[b * 16 + 2]) * 0.1 / grid;
        }
        fprintf(stderr, "[mlp_co hidden=%d grid=%d tiles=%d(+%d)] total %.1f us | tile %d of each workgroup: rows+LN %.1f, chunks %.1f (%.2f GHz), epilogue %.1f\n",
                p.hidden, grid, p.full_tiles, p.ntiles - p.full_tiles, (double)(t4 - t0) * 0.01, PSTAMP_SEQ, pro, chunks, ghz, epi);
    }
#endif
    return HIPT_OK;
}
