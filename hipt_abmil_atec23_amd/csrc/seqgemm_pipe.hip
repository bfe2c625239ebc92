// A-STATIONARY GEMM, software-pipelined, for K = 384 and N a multiple of 128 (bf16 mode):
//   out[M, N] (bf16) = LN?(x)[M, 384] @ W[N, 384]^T + bias
// QKV with LayerNorm-1 fused (HIPT_4K/vision_transformer.py:114,121,147) and attn.proj (:116,129; the branch
// output stays bf16, the residual add is folded into the consumer).  Same arithmetic as seqgemm.hip; what changes:
//   * 192-row tiles (3 MFMA row fragments per wave): the operand fragments take 144 VGPRs, which leaves the
//     accumulator file room for TWO [192, 128] accumulators.  While the MFMAs of N tile t+1 run, the vector
//     and store slots under them convert and store N tile t: the epilogue disappears from the critical path
//     (it was 30 % of the kernel: skipping its stores took 1109 -> 760 us at 8 regions).
//   * the weight rows of a slab are laid into LDS in a permuted column order (the DMA picks the global row)
//     such that a lane's two fragments of a pair hold 8 CONSECUTIVE output columns: 16-byte stores, not 8.
//   * stores go through a per-tile buffer resource: rows past the tile end are out of range for the hardware,
//     no predication (a branch would cut the scheduling region around the MFMAs).
//   * one ring unit = 3 slabs (48 KiB), 3 units in LDS, one barrier per unit, fragment reads one 12-MFMA group
//     ahead across unit boundaries, bias as the C operand of an accumulator's first MFMA; persistent workgroups
//     pull row tiles from an atomic counter and the weight stream runs continuously across tiles.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "pipe_common.h"

namespace {

constexpr int K = 384, NCH = 12, TMR = 192, MF = 3;
constexpr int SLAB = 16384, UNIT = 3 * SLAB;
constexpr int MAXN = 2048;
constexpr int PIPE_LDS = 3 * UNIT + 2 * K * 4 + MAXN * 4 + 16;

#define QSTAMP(k)                                                                                                    \
    do {                                                                                                             \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == 0) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

typedef std::integral_constant<int, 0> I0;
typedef std::integral_constant<int, 1> I1;
typedef std::integral_constant<int, -1> IM1;

// DBG (tools/seqgemm_probe.hip only): 1 = no weight DMA / ring syncs, 2 = no output stores, 4 = no LDS reads / MFMAs
// per-lane part of a weight DMA address: LDS row R = wave*8 + (lane>>3) + 32q of a slab takes output column n0 + U(R),
// U(16nf + 4g' + e) = 32(nf>>1) + 8g' + 4(nf&1) + e (lane g' then owns, over the fragment pair (2f, 2f+1), output columns
// 32f + 8g' + (0..7); U(R + 32) = U(R) + 32: piece q only adds a uniform offset); the lane's 16-byte chunk is its LDS
// chunk position XOR ((R>>1)&7)
__device__ __forceinline__ uint32_t lane_src(int wave, int lane) {
    const int r0 = wave * 8 + (lane >> 3);
    const int ch0 = (lane & 7) ^ ((r0 >> 1) & 7);
    const int urow = 8 * ((r0 >> 2) & 3) + 4 * ((r0 >> 4) & 1) + (r0 & 3);
    return (uint32_t)(urow * K + ch0 * 8) * 2;
}
// byte offset in W of piece (j, q) of ring unit (nt, kh), without the lane part
__device__ __forceinline__ int64_t piece_src(int nt, int kh, int j, int q) { return ((int64_t)nt * 128 * K + kh * 192 + j * 64 + q * 32 * K) * 2; }

// one thread per 16-byte chunk of the packed image
__global__ void seqgemm_pack_kernel(const char* __restrict__ W, int N, char* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)(N >> 7) * 2 * (UNIT / 16)) return;
    const int u = (int)(i / (UNIT / 16)), o = (int)(i % (UNIT / 16)) * 16;
    const int j = o / SLAB, pw = (o % SLAB) / 1024, q = pw >> 2, wave = pw & 3, lane = (o % 1024) / 16;
    *(u32x4*)(out + (int64_t)u * UNIT + o) = *(const u32x4*)(W + piece_src(u >> 1, u & 1, j, q) + lane_src(wave, lane));
}

// PACKED: the weights come from the pre-packed image p.wpk (a DMA piece = 1 KiB of consecutive bytes) instead of p.W
// AIMG: A (bf16, LN = false) is an activation image;  OIMG: the output (N = 384 = ldc) is written as one (kernels.h)
// OHM: the output (N = 3 * 384: q | k | v of 6 heads x 64) is written head-major, [sequence][q/k/v][head][token][64]
//      (p.out_ntok tokens per sequence): the attention kernel then stages a head's K / V with consecutive 1 KiB pieces
template <bool LN, int DBG = 0, bool PACKED = false, bool AIMG = false, bool OIMG = false, bool OHM = false>
__global__ __launch_bounds__(256, 1) void seqgemm_pipe_kernel(const SeqGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gam = (float*)(smem + 3 * UNIT);
    float* bet = gam + K;
    float* bia = bet + K;                 // [N], permuted like the weight rows
    int* tile_s = (int*)(bia + MAXN);     // [2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int NT = p.N >> 7;    // 128-column tiles
    const int upt = 2 * NT;     // ring units per pass over the weights
    const int rot = blockIdx.x % NT;  // this workgroup's first N tile: spreads the stores of the chip over all columns

    // ---- weight DMA (address parts: lane_src / piece_src above) ----
    const uint32_t loff = PACKED ? (uint32_t)(wave * 1024 + lane * 16) : lane_src(wave, lane);  // everything else is wave-uniform
    const char* ibase = (const char*)(PACKED ? p.wpk : p.W);
    int islot = 0, ipos = 0;
    auto set_issue = [&](int pos, int slot) {  // unit pos of a pass: N tile (rot + pos/2) % NT, k slabs 3(pos&1) ..
        int nt = rot + (pos >> 1);
        nt = nt >= NT ? nt - NT : nt;
        ibase = PACKED ? (const char*)p.wpk + (int64_t)(2 * nt + (pos & 1)) * UNIT : (const char*)p.W + piece_src(nt, pos & 1, 0, 0);
        islot = slot;
    };
    auto dma_piece = [&](auto T_) __attribute__((always_inline)) {
        constexpr int t = decltype(T_)::value, j = t >> 2, q = t & 3;
        if constexpr ((DBG & 1) == 0)
            glds16(ibase + (PACKED ? j * SLAB + q * 4096 : j * 64 * 2 + q * 32 * K * 2) + loff, smem + islot * UNIT + j * SLAB + (q * 4 + wave) * 1024);
    };

    for (int i = tid; i < p.N; i += 256)
        bia[i] = p.bias ? p.bias[(i & ~31) + 8 * ((i >> 2) & 3) + 4 * ((i >> 4) & 1) + (i & 3)] : 0.f;
    if constexpr (LN) {
        for (int i = tid; i < K; i += 256) {
            gam[i] = p.ln_w[i];
            bet[i] = p.ln_b[i];
        }
    }
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

    uint32_t foff[2];
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1) foff[k1] = li * 128 + (((g + 4 * k1) ^ ((lane >> 1) & 7)) << 4);
    const uint32_t lbase = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    const uint32_t biabase = (uint32_t)(uintptr_t)(LDS_AS char*)bia + 16 * g;
    const uint32_t tsbase = (uint32_t)(uintptr_t)(LDS_AS char*)tile_s;
    const uint32_t gbase = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 32 * g;
    // store offsets inside a tile's buffer: row (wave*3 + mf)*16 + li, column 8g (+ uniform n0 + 32f)
    int voff[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)  // (image: fragment (wave*3 + mf) of the tile, chunk position lane; + column chunk * 1 KiB)
        voff[mf] = OIMG ? (wave * MF + mf) * 16 * K * 2 + lane * 16 : (int)((((wave * MF + mf) * 16 + li) * p.ldc + 8 * g) * 2);

    // ---- prime the ring: unit 0 whole, the first two pieces of unit 1 ----
    int cons = 0;
    if (tile < p.ntiles) {
        set_issue(0, 0);
        sfor<0, 12>(dma_piece);
        set_issue(1, 1);
        dma_piece(I0{});
        dma_piece(I1{});
        ipos = 2 % upt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    u32x4 wA[2][4];
    auto rd_frag = [&](auto SET_, auto G_, uint32_t sa) __attribute__((always_inline)) {
        // group gg of a unit: slab j = gg>>2, k-step ks = (gg>>1)&1, column fragments 4(gg&1) + n -> LDS rows
        // 16(4(gg&1) + n) + li, chunk 4ks + g
        constexpr int set = decltype(SET_)::value, gg = decltype(G_)::value, j = gg >> 2, k1 = (gg >> 1) & 1;
        constexpr int off = j * SLAB + (gg & 1) * 8192;
        if constexpr ((DBG & 4) == 0) {
            const uint32_t a = sa + foff[k1];
            u32x4 &d0 = wA[set][0], &d1 = wA[set][1], &d2 = wA[set][2], &d3 = wA[set][3];
            DSR128(d0, a, off);
            DSR128(d1, a, off + 2048);
            DSR128(d2, a, off + 4096);
            DSR128(d3, a, off + 6144);
        }
    };
    f32x4 bq[4];  // bias of the 4 column fragments an accumulator quad starts from
    auto bias_rd = [&](int off) __attribute__((always_inline)) {  // off: float offset of the quad inside bia
        const uint32_t a = biabase + off * 4;
        f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
        DSR128(q0, a, 0);
        DSR128(q1, a, 64);
        DSR128(q2, a, 128);
        DSR128(q3, a, 192);
    };

    for (int seq = 0; tile < p.ntiles; ++seq) {
        const int row0 = tile * TMR;
        int nrows = p.M - row0;
        nrows = nrows < TMR ? nrows : TMR;
        QSTAMP(0);
        if (tid == 0) {  // next tile: fetched now, read after this tile's ring barriers
            const int nt = atomicAdd(p.counter, 1);
            asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt) : "memory");
            if (nt == last_fetch) *p.counter = 0;
        }
        // this tile's output window: rows beyond nrows are outside the buffer -> their stores are dropped
        const __amdgpu_buffer_rsrc_t orsrc =
            OHM ? __builtin_amdgcn_make_buffer_rsrc((char*)p.out, 0, (int)(uint32_t)((int64_t)p.M * p.N * 2), 0x00020000)
                : __builtin_amdgcn_make_buffer_rsrc((char*)p.out + (int64_t)row0 * p.ldc * 2, 0, (int)((int64_t)nrows * p.ldc * 2), 0x00020000);
        if constexpr (OHM) {  // row -> (sequence b, token t): byte offset of element (b, q, head 0, t, 8g); rows past M: out of range
            const float inv = 1.0f / (float)p.out_ntok;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int row = row0 + (wave * MF + mf) * 16 + li;
                int b = (int)(((float)row + 0.5f) * inv);
                b = b * p.out_ntok > row ? b - 1 : b;
                b = (b + 1) * p.out_ntok <= row ? b + 1 : b;
                const int t = row - b * p.out_ntok;
                voff[mf] = row < p.M ? ((b * 18 * p.out_ntok + t) * 64 + 8 * g) * 2 : (int)0xfffffff0u;
            }
        }

        // ---- activations -> operand fragments ----
        u32x4 af[MF][NCH];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            int r = (wave * MF + mf) * 16 + li;
            r = r < nrows ? r : nrows - 1;  // rows past the end re-read the last valid row (never stored)
            if constexpr (LN) {
                const float* xr = (const float*)p.A + (int64_t)(row0 + r) * p.lda;
                f32x4 v[NCH][2];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    v[c][0] = *(const f32x4*)(xr + (g + 4 * c) * 8);
                    v[c][1] = *(const f32x4*)(xr + (g + 4 * c) * 8 + 4);
                }
                ln_rows_lds<K, NCH>(v, gbase, p.ln_eps, af[mf]);
                // park the finished fragment in the accumulator file (idle during the row phase) while the next one
                // is loaded and normalised: left alone, hipcc sends finished fragments to scratch instead and the
                // reloads stall the first N tile (measured: +23 us)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    u32x4& a2 = af[mf][c];
                    asm volatile("" : "+a"(a2));
                }
                __builtin_amdgcn_sched_barrier(0);
            } else {
                if constexpr (AIMG) {  // whole fragments (M % 16 == 0); one past the tile's end re-reads fragment 0 (never stored)
                    const int fr = (wave * MF + mf) * 16 < nrows ? (wave * MF + mf) * 16 : 0;
                    const bf16_t* ar = (const bf16_t*)p.A + (int64_t)(row0 + fr) * K + lane * 8;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) af[mf][c] = *(const u32x4*)(ar + c * 512);
                } else {
                    const bf16_t* ar = (const bf16_t*)p.A + (int64_t)(row0 + r) * p.lda;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) af[mf][c] = *(const u32x4*)(ar + (g + 4 * c) * 8);
                }
            }
        }
        if constexpr (LN) {
            // Register budget: 256 VGPRs + 256 accumulator registers, 192 of the latter taken by the two accumulators.
            // The last fragment's operands live in the remaining accumulator registers (an MFMA reads its A/B operands
            // from either file): 96 operand VGPRs instead of 144, and the LayerNorm above fits without scratch.
#pragma unroll
            for (int mf = 0; mf < MF - 1; ++mf)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    u32x4& a2 = af[mf][c];
                    asm volatile("" : "+v"(a2));
                }
        }
        QSTAMP(2);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == 0) p.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memtime();

        f32x4 acc[2][MF][8];

        // one 16-byte store: fragment pair f of row fragment mf of the N tile whose first column is n0e
        auto store_unit = [&](auto BUF_, auto U_, int n0e) __attribute__((always_inline)) {
            constexpr int buf = decltype(BUF_)::value, u = decltype(U_)::value, mf = u >> 2, f = u & 3;
            u32x4 o;
            o[0] = pack_bf16x2(acc[buf][mf][2 * f][0], acc[buf][mf][2 * f][1]);
            o[1] = pack_bf16x2(acc[buf][mf][2 * f][2], acc[buf][mf][2 * f][3]);
            o[2] = pack_bf16x2(acc[buf][mf][2 * f + 1][0], acc[buf][mf][2 * f + 1][1]);
            o[3] = pack_bf16x2(acc[buf][mf][2 * f + 1][2], acc[buf][mf][2 * f + 1][3]);
            // (image: column chunk (n0e + 32f) / 32 of the fragment, 1 KiB each)
            if constexpr (OHM) {  // columns n0e + 32f + 8g ..+7 lie in one head: (q/k/v, head, d0) -> uniform byte offset
                const int col0 = n0e + 32 * f, which = col0 >= 2 * K ? 2 : (col0 >= K ? 1 : 0), rem = col0 - which * K;
                const int soff = (((which * 6 + (rem >> 6)) * p.out_ntok) * 64 + (rem & 63)) * 2;
                if constexpr ((DBG & 2) == 0) __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, voff[mf], soff, 0);
            } else {
                if constexpr ((DBG & 2) == 0) __builtin_amdgcn_raw_buffer_store_b128(o, orsrc, voff[mf], (n0e + 32 * f) * (OIMG ? 32 : 2), 0);
            }
        };

        // ---- one phase = one ring unit = 3 k slabs of one N tile: 12 groups of 12 MFMAs ----
        // BUF: accumulator of this N tile;  HALF: k slabs 3 HALF ..;  EPI: the OTHER accumulator (N tile n0e) is
        // converted and stored under this phase, store units 6 EPI .. 6 EPI + 5 (EPI = -1: nothing to store)
        // NB: 1 = the next phase starts an N tile (bias quad at float offset nb is read ahead), 0 = it does not,
        //     -1 = last phase of the row tile: nothing is prefetched
        auto phase = [&](auto BUF_, auto HALF_, auto EPI_, auto NB_, int nb, int nb2, int n0e) __attribute__((always_inline)) {
            constexpr int buf = decltype(BUF_)::value, half = decltype(HALF_)::value, epi = decltype(EPI_)::value;
            constexpr int needb = decltype(NB_)::value;
            const uint32_t sa = lbase + (cons % 3) * UNIT;
            const uint32_t sn = lbase + ((cons + 1) % 3) * UNIT;
            sfor<0, 12>([&](auto G_) __attribute__((always_inline)) {
                constexpr int gg = decltype(G_)::value, set = gg & 1;
                typedef std::integral_constant<int, set ^ 1> NS;
                if constexpr (gg == 11) {
                    if constexpr ((DBG & 1) == 0) {
                        // my pieces of the next unit have landed (the 6 stores of this phase are younger: they may fly)
                        if constexpr (epi >= 0 && (DBG & 2) == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();  // ... everyone's; unit cons-1 is no longer read
                        set_issue(ipos, (cons + 2) % 3);
                        ipos = ipos + 1 == upt ? 0 : ipos + 1;
                    }
                    if constexpr (needb < 0) {
                        LGKM(0);
                    } else {
                        rd_frag(NS{}, I0{}, sn);
                        if constexpr (needb > 0) {
                            bias_rd(nb);
                            LGKM(8);
                        } else {
                            LGKM(4);
                        }
                    }
                } else {
                    rd_frag(NS{}, std::integral_constant<int, gg + 1>{}, sa);
                    if constexpr (half == 0 && gg == 0) {  // second bias quad of this N tile, for group 1
                        // (group 0's quad moves to the accumulators first: its registers are free again)
                        LGKM(4);
                    } else {
                        LGKM(4);
                    }
                }
                if constexpr ((DBG & 4) == 0) {
                    constexpr int j = gg >> 2, ks = (gg >> 1) & 1, nfq = gg & 1, kidx = 2 * (3 * half + j) + ks;
                    if constexpr (half == 0 && gg < 2) {
                        // first MFMAs of these 4 column fragments: C = bias (read one group / one phase ago, landed NOW)
                        f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
                        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int mf = 0; mf < MF; ++mf) {
                            if constexpr (half == 0 && gg < 2) {
                                f32x4 t = bq[n];
                                Tr<bf16_t>::mma16(t, wA[set][n], af[mf][kidx]);
                                acc[buf][mf][4 * nfq + n] = t;
                            } else {
                                Tr<bf16_t>::mma16(acc[buf][mf][4 * nfq + n], wA[set][n], af[mf][kidx]);
                            }
                        }
                    if constexpr (half == 0 && gg == 0) {
                        // group 1 starts the other 4 column fragments: their bias goes into the same registers, after
                        // the MFMAs above have taken theirs (program order; the wait of group 1 covers the landing)
                        __builtin_amdgcn_sched_barrier(0);
                        bias_rd(nb2);
                    }
                }
                if constexpr (gg == 11) {
                    dma_piece(I0{});
                    dma_piece(I1{});
                } else if constexpr (gg <= 4) {
                    dma_piece(std::integral_constant<int, 2 + 2 * gg>{});
                    dma_piece(std::integral_constant<int, 3 + 2 * gg>{});
                }
                if constexpr (epi >= 0 && gg >= 5 && gg <= 10)
                    store_unit(std::integral_constant<int, buf ^ 1>{}, std::integral_constant<int, 6 * (epi >= 0 ? epi : 0) + gg - 5>{}, n0e);
                __builtin_amdgcn_sched_barrier(0);
            });
            cons += 1;
        };
        // N tile index t of this pass -> first column
        auto ncol = [&](int t) {
            int nt = rot + t;
            nt = nt >= NT ? nt - NT : nt;
            return nt * 128;
        };

        // first fragments and bias of the pass (asm reads land asynchronously: nothing but the first phase may sit
        // between them and their counted wait)
        rd_frag(I0{}, I0{}, lbase + (cons % 3) * UNIT);
        bias_rd(ncol(0));
        // N tile 0 (nothing to store yet), then pairs (accumulators 1, 0), then a possible odd one
        phase(I0{}, I0{}, IM1{}, I0{}, 0, ncol(0) + 64, 0);
        phase(I0{}, I1{}, IM1{}, I1{}, ncol(1 < NT ? 1 : 0), 0, 0);
        QSTAMP(5);
        int t = 1;
        for (; t + 1 < NT; t += 2) {
            phase(I1{}, I0{}, I0{}, I0{}, 0, ncol(t) + 64, ncol(t - 1));
            phase(I1{}, I1{}, I1{}, I1{}, ncol(t + 1), 0, ncol(t - 1));
            phase(I0{}, I0{}, I0{}, I0{}, 0, ncol(t + 1) + 64, ncol(t));
            phase(I0{}, I1{}, I1{}, I1{}, ncol(t + 2 < NT ? t + 2 : 0), 0, ncol(t));
        }
        LGKM(0);  // (the last phase read fragments / a bias quad nobody uses: let them land before their registers are
        {         //  re-used, and keep those registers allocated up to here: a fake use AFTER the wait)
            f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
            u32x4 &w0 = wA[0][0], &w1 = wA[0][1], &w2 = wA[0][2], &w3 = wA[0][3];
            asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(w0), "v"(w1), "v"(w2), "v"(w3));
        }
        int nlast;
        if (t < NT) {  // odd one left: accumulator 1
            rd_frag(I0{}, I0{}, lbase + (cons % 3) * UNIT);
            bias_rd(ncol(t));
            phase(I1{}, I0{}, I0{}, I0{}, 0, ncol(t) + 64, ncol(t - 1));
            phase(I1{}, I1{}, I1{}, IM1{}, 0, 0, ncol(t - 1));
            nlast = ncol(t);
            sfor<0, 12>([&](auto U_) __attribute__((always_inline)) { store_unit(I1{}, U_, nlast); });
        } else {
            nlast = ncol(NT - 1);
            sfor<0, 12>([&](auto U_) __attribute__((always_inline)) { store_unit(I0{}, U_, nlast); });
        }
        __builtin_amdgcn_sched_barrier(0);
        QSTAMP(3);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == 0) p.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime();
        QSTAMP(4);
        if (DBG & 1) __syncthreads();  // (no ring barriers in this debug build)
        int nt;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nt) : "v"(tsbase + 4 * ((seq + 1) & 1)) : "memory");
        tile = __builtin_amdgcn_readfirstlane(nt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: pieces of a pass that never runs)
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

bool hipt_seqgemm_pipe_supported(int dtype, int K_, int N, bool ln, int flags) {
    return dtype == HIPT_BF16 && K_ == 384 && N % 128 == 0 && N >= 256 && N <= MAXN && flags == 0;
}

int hipt_seqgemm_pack_launch(const void* W, int N, int K_, void* packed, hipStream_t st) {
    if (!hipt_seqgemm_pipe_supported(HIPT_BF16, K_, N, false, 0)) {
        hipt_set_error("seqgemm pack: unsupported K=%d N=%d", K_, N);
        return HIPT_E_UNSUPPORTED;
    }
    const int64_t chunks = (int64_t)(N >> 7) * 2 * (UNIT / 16);
    hipLaunchKernelGGL(seqgemm_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const char*)W, N, (char*)packed);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

template <bool LN, int DBG>
int hipt_seqgemm_pipe_launch_dbg(const SeqGemmParams& p_in, hipStream_t st) {
    SeqGemmParams p = p_in;
    // p.img: bit 0 = A is an activation image (LN = false), bit 1 = the output is written as one (N = ldc = 384); both
    // need the packed weights (one instantiation less) and whole 16-row fragments
    // bit 2 = head-major q|k|v output (N = 1152 = ldc, p.out_ntok set, the buffer under 4 GiB)
    if (p.img && (((p.img & 3) && (LN || p.M % 16 != 0)) || p.img > 5 || p.img == 3 + 4 || ((p.img & 2) && (p.N != K || p.ldc != K)) ||
                  ((p.img & 1) && p.lda != K) || ((p.img & 4) && (p.N != 3 * K || p.ldc != 3 * K || p.out_ntok <= 0 || (p.img & 2) ||
                                                                (int64_t)p.M * p.N * 2 >= ((int64_t)1 << 32) - 65536)))) {
        hipt_set_error("seqgemm_pipe: activation images: unsupported combination (img=%d LN=%d M=%d N=%d lda=%lld ldc=%lld packed=%d)", p.img,
                       (int)LN, p.M, p.N, (long long)p.lda, (long long)p.ldc, p.wpk != nullptr);
        return HIPT_E_BADARG;
    }
    if (!p.wpk) {
        hipt_set_error("seqgemm_pipe: needs the packed weight image (hipt_seqgemm_pack_launch)");
        return HIPT_E_BADARG;
    }
    auto k = seqgemm_pipe_kernel<LN, DBG, true>;
    if constexpr (LN) {
        if (p.img == 4) k = seqgemm_pipe_kernel<true, DBG, true, false, false, true>;
    }
    if constexpr (!LN) {
        if (p.img == 5) k = seqgemm_pipe_kernel<false, DBG, true, true, false, true>;
        if (p.img == 4) k = seqgemm_pipe_kernel<false, DBG, true, false, false, true>;
        if (p.img == 1) k = seqgemm_pipe_kernel<false, DBG, true, true, false>;
        if (p.img == 2) k = seqgemm_pipe_kernel<false, DBG, true, false, true>;
        if (p.img == 3) k = seqgemm_pipe_kernel<false, DBG, true, true, true>;
    }
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)seqgemm_pipe_kernel<LN, DBG, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)seqgemm_pipe_kernel<false, DBG, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)seqgemm_pipe_kernel<false, DBG, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)seqgemm_pipe_kernel<false, DBG, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)seqgemm_pipe_kernel<false, DBG, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)seqgemm_pipe_kernel<LN, DBG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(seqgemm_pipe) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("seqgemm_pipe: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    const int ncu = once.ncu[dev];
    p.ntiles = (p.M + TMR - 1) / TMR;
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    if (!p.counter_zeroed && hipMemsetAsync(p.counter, 0, sizeof(int), st) != hipSuccess) {
        hipt_set_error("seqgemm_pipe: hipMemsetAsync(counter) failed");
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
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), PIPE_LDS, st, p);
    HIPT_CHECK_LAUNCH();
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
        double pro = 0, first = 0, rest = 0, ghz = 0;
        for (int b = 0; b < grid; ++b) {
            pro += (double)(h[b * 16 + 2] - h[b * 16 + 0]) * 0.01 / grid;
            first += (double)(h[b * 16 + 5] - h[b * 16 + 2]) * 0.01 / grid;
            rest += (double)(h[b * 16 + 3] - h[b * 16 + 5]) * 0.01 / grid;
            ghz += (double)(h[b * 16 + 9] - h[b * 16 + 8]) / (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.1 / grid;
        }
        fprintf(stderr, "[seqgemm_pipe LN=%d dbg=%d N=%d grid=%d tiles=%d] total %.1f us | first tiles: rows%s %.1f, N tile 0 %.1f, N tiles 1.. + last stores %.1f (%.2f GHz)\n",
                (int)LN, DBG, p.N, grid, p.ntiles, (double)(t4 - t0) * 0.01, LN ? "+LN" : "", pro, first, rest, ghz);
    }
#endif
    return HIPT_OK;
}

int hipt_seqgemm_pipe_launch(const SeqGemmParams& p, bool ln, hipStream_t st) {
    HIPT_CHECK_ARG(p.counter != nullptr, "seqgemm_pipe: null tile counter");
    return ln ? hipt_seqgemm_pipe_launch_dbg<true, 0>(p, st) : hipt_seqgemm_pipe_launch_dbg<false, 0>(p, st);
}
