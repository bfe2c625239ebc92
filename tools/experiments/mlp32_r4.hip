// FUSED MLP SUB-BLOCK for D = 384 (ViT-256) on 32x32x16 MFMAs:   x <- x + y1 + fc2( GELU( fc1( LN2(x + y1) ) ) )
//   (Block.forward second half, HIPT_4K/vision_transformer.py:151 with Mlp.forward :98-104.)
//
// One 4-wave workgroup owns 128 rows; every wave its 32 rows end to end, all in registers at one wave per SIMD: LN2(x + y1) as MFMA
// operand fragments, fc1 half-chunk accumulators -> GELU -> re-packed in registers as the fc2 operand, the [32, 384] fc2 accumulator.
// Only weights stream through a 3 x 48 KiB LDS-DMA ring; phases A0(c) B1(c-1) A1(c) B0(c) so that a half's GELU hides under the two
// phases that follow its fc1.
//   * v_mfma_f32_32x32x16_bf16.  The kernel is ISSUE-bound, not MFMA-bound: per 128 MFMA cycles a wave also has to issue ~70 cycles
//     of GELU arithmetic, ~70 of LDS-DMA pieces and its fragment reads; a 32x32x16 MFMA holds the SIMD's vector issue for 8 of its
//     32 cycles (96 of 128 left; a 16x16x32 for 8 of 16).
//   * a wave's 32 rows are ONE B operand (column = row): lane l = 32 h + 16 m + li holds row (fragment m, li) and, per 16-deep
//     k-step, 8 k values of half h.
//   * round 4: the row phase works in the ACCUMULATOR layout.  A lane loads its row's x and y1 as the 16-byte / 8-byte pieces the
//     fc2 accumulator tiles hold (columns 32 O + 8 q + 4 h + e), takes LayerNorm-2 there (the two h-lanes of a row hold all of it),
//     packs the normalised values straight into the fc1 operand -- k-step 2 O + p of lane half h carries columns 32 O + 16 p +
//     8 (j >> 2) + 4 h + (j & 3), the order the weight image is built for -- and SEEDS the fc2 accumulators with v = x + y1.  The
//     epilogue adds b2, stores, and applies the next block's LayerNorm-1: x and y1 are read ONCE per launch (round 3 re-read them
//     in the epilogue: 1.21 of 3.99 GB per 8-region launch; tools/experiments/mlp32_r3.hip keeps that kernel for A/B runs).
//   * weights as A operand: one fragment = 32 output units x 16 k = 1 KiB = one ds_read_b128 per lane; the packed image stores
//     the fragments of a ring unit in consumption order, each as 64 x 16 consecutive bytes: every LDS read is conflict-free by
//     construction, every DMA piece is 1 KiB of consecutive bytes.
//   * fc1 accumulator tile (32 hidden x 32 rows): lane holds its row's hidden units (reg & 3) + 8 (reg >> 2) + 4 h.  After GELU,
//     registers 8 s .. 8 s + 7 packed to bf16 ARE the fc2 operand fragment of k-step s (accumulator-as-operand); the fc2
//     weight image lists the hidden units in that order.
//   * fc2 accumulator tile (32 outputs x 32 rows): lane holds 4 runs of 4 consecutive output columns 32 O + 8 q + 4 h + (0..3):
//     16-byte pieces of the fp32 row, 8-byte pieces of the bf16 ones.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "mlp_common.h"
#include "pipe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int D = 384, NKS = 24, NOT = 12, TMR = 128;   // NKS: 16-deep k-steps of fc1; NOT: 32-wide output tiles
constexpr int UNIT = 48 * 1024;                                    // ring unit = one phase = 48 fragments of 1 KiB

template <int DBG = 0>
__device__ __forceinline__ void mma32(f32x16& acc, const u32x4& a, const u32x4& b) {
    if constexpr (DBG & 4) {
        asm volatile("" : "+v"(acc) : "v"(a), "v"(b));
        return;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

#ifndef PSTAMP_SEQ
#define PSTAMP_SEQ 0  // which tile of a workgroup the debug stamps describe (0 = the first: every CU in step)
#endif
// Diagnostic builds only.  A stamp is read into SCALAR registers where it happens, unconditionally, and a tile's stamps are stored in
// one place at the end of the tile: even a uniform branch around a clock read is a basic-block boundary, and one between the row
// phase and the first ring phase makes hipcc spill the operand registers (353 spills in such a stamp build against 2 in the release
// build) -- the stamp build would time a different kernel.
#ifdef HIPT_DEBUG_STAMPS
#define PSTAMP(k) stamp_rt[k] = __builtin_amdgcn_s_memrealtime()
#define PSTAMP_CLK(k) stamp_clk[k] = __builtin_amdgcn_s_memtime()
#else
#define PSTAMP(k) (void)stamp_rt
#define PSTAMP_CLK(k) (void)stamp_clk
#endif

enum { KA = 0, KB = 1 };  // phase kind: fc1 half / fc2 half

// Ring unit `pos` of a tile pass: positions A0(0) A1(0) B0(0) | A0(c) B1(c-1) A1(c) B0(c) ... | B1(n-1)
__device__ __forceinline__ void unit_of(int pos, int nchunk, bool& is_a, int& c, int& h) {
    const int upt = 4 * nchunk;
    if (pos < 3) {
        c = 0;
        is_a = pos < 2;
        h = pos == 1 ? 1 : 0;
    } else if (pos == upt - 1) {
        c = nchunk - 1;
        is_a = false;
        h = 1;
    } else {
        const int m = pos - 3, r = m & 3;
        c = 1 + (m >> 2);
        is_a = (r & 1) == 0;
        h = r == 2 ? 1 : (r == 1 ? 1 : 0);
        if (r == 1) c -= 1;
    }
}

// The packed image: unit after unit in pass order, each 48 fragments x 1 KiB, lane-major (lane l = 32 h + r: 16 bytes at l * 16).
//   fc1 unit (chunk c, half hh: hidden Hb = 128 c + 64 hh): fragment 4 gg + 2 p + U (gg 0..11, p 0/1, tile U 0/1): element j =
//       W1[Hb + 32 U + r][32 gg + 16 p + 8 (j >> 2) + 4 h + (j & 3)]        (k-step s = 2 gg + p of the activations' k order: the
//       column order of an ACCUMULATOR tile, so that an operand may also come straight from one -- the fc2 order, see below)
//   fc2 unit: fragment 4 O + t (output tile O 0..11, t = 2 U + s'): element j of lane (r, h) =
//       W2[32 O + r][Hb + 32 U + 16 s' + 8 (j >> 2) + 4 h + (j & 3)]         (the hidden order of a GELU'd fc1 accumulator tile)
__global__ void mlp32_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, int hidden, u32x4* __restrict__ out) {
    const int nchunk = hidden / 128, upt = 4 * nchunk;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte lane chunk
    if (i >= (int64_t)upt * (UNIT / 16)) return;
    const int pos = (int)(i / (UNIT / 16)), o = (int)(i % (UNIT / 16)), frag = o >> 6, lane = o & 63, r = lane & 31, h = lane >> 5;
    bool is_a;
    int c, hh;
    unit_of(pos, nchunk, is_a, c, hh);
    const int Hb = 128 * c + 64 * hh;
    if (is_a) {
        const int gg = frag >> 2, pp = (frag >> 1) & 1, U = frag & 1;
        const bf16_t* row = w1 + (int64_t)(Hb + 32 * U + r) * D + 32 * gg + 16 * pp + 4 * h;
        const u32x2 lo = *(const u32x2*)row, hi = *(const u32x2*)(row + 8);
        out[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
        const int O = frag >> 2, t = frag & 3, U = t >> 1, s2 = t & 1;
        const bf16_t* row = w2 + (int64_t)(32 * O + r) * hidden + Hb + 32 * U + 16 * s2 + 4 * h;
        const u32x2 lo = *(const u32x2*)row, hi = *(const u32x2*)(row + 8);
        out[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
}

// IMG / XIN: fragment-blocked activation images (kernels.h, "activation images") -- IMG: y1 is read and x / xn_out are written as
// images; XIN: x is read as an image.  Row-major otherwise.  Weights always come from the packed image p.wpk (format 1).
// DBG (tools/mlp_probe.hip only): 1 = no weight DMA / ring syncs, 2 = GELU replaced by a plain pack, 4 = no MFMAs, 8 = no LDS
// fragment reads.
template <bool IMG = false, bool XIN = false, int DBG = 0>
__global__ __launch_bounds__(256, 1) void mlp32_kernel(const MlpParams p) {
    // RING2 (round 4): TWO weight units in flight.  Rounds 1-3 requested unit c + 1 in groups 0..4 of phase c and waited for it at the
    // end of the same phase: the last pieces had 7 groups (~0.8 us) to come from L2 -- under the load of 256 CUs streaming the image
    // that is the L2 -> LDS latency itself, every phase ended in that wait, and whatever the waves did in between (MFMA shape, GELU
    // form, LDS reads, row prefetches) landed on the same time; the ring's third slot held a dead unit all the while.  Now phase c
    // requests unit c + 2 into that slot and its closing wait is vmcnt(12): a unit has a whole phase more to land.
    // (DBG 128, tools/mlp_probe.hip: the old protocol)
    constexpr bool RING2 = (DBG & 128) == 0;
    constexpr bool SGB = (DBG & 256) == 0;  // (DBG 256: hipcc's own instruction order inside the groups)
    // DBG 64 (tools/mlp_probe.hip): WITH the L2 prefetch of the next tile's rows (prefetch_rows below).  Measured, interleaved on one
    // box at 8 regions: 1 419-1 432 us with it, 1 384-1 388 without -- the row phase falls from 13.5 to 9 us per tile, the chunk phases
    // take 6 k more cycles and the chip gives the saved idle time back as clock (1.63 instead of 1.73 GHz): off.
    constexpr bool PF = (DBG & 64) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gam = (float*)(smem + 3 * UNIT);
    float* bet = gam + D;
    float* b2s = bet + D;
    float* b1s = b2s + D;                  // [hidden]
    int* tile_s = (int*)(b1s + p.hidden);  // [2] tile handed to this workgroup, double-buffered by parity
    float* gam1 = (float*)(tile_s + 4);    // next block's LayerNorm-1 (gamma | beta), if p.xn_out
    float* pfj = gam1 + 2 * D;             // [64] where the L2-prefetch loads below drop their dwords (never read)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = p.hidden / 128;
    const int upt = 4 * nchunk;                   // ring units (phases) per tile pass

    // ---- weight DMA: unit pos of the image = 48 pieces of 1 KiB, byte for byte what its ring slot holds; wave w issues pieces
    // 12 w .. 12 w + 11.  An LDS-DMA instruction takes its LDS base from M0, and it is WRITING M0 that makes a piece expensive
    // (tools/issue_mix_probe.hip: +36 cycles per piece with a new M0, +2 with the same M0 and the piece selected by the instruction's
    // immediate offset, which is added to the LDS and to the global address alike): four consecutive pieces share one M0.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 2 * p.hidden * D * 2, 0x00020000);
    const uint32_t ilane = (uint32_t)(12 * wave * 1024 + lane * 16);
    int ioff = 0, islot = 0, ipos = 0;
    auto set_issue = [&](int pos, int slot) {
        ioff = pos * UNIT;
        islot = slot;
    };
    auto dma_piece = [&](auto T_) __attribute__((always_inline)) {
        constexpr int t = decltype(T_)::value;
        if constexpr ((DBG & 1) == 0 && (DBG & 32) == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + islot * UNIT + (12 * wave + (t & ~3)) * 1024), 16, ilane, ioff + (t & ~3) * 1024, (t & 3) * 1024, 0);
    };

    // ---- L2 prefetch of the NEXT tile's row-phase inputs.  The row phase is latency: with 255 other CUs streaming weights a wave waits
    // ~9 us for its 36 KiB of rows (14 us row phase; 6.7 us when the chip is otherwise quiet).  Touching one dword of every 128-byte
    // line of the next tile's x and y1 rows from inside the LAST ring phase of this tile turns that wait into L2 hits.  The loads
    // are LDS-DMA (no destination register: nothing for the register allocator to keep alive), all into one 256-byte scratch line;
    // rows past the tile's end are out of the resource's range and dropped.  Nine instructions per wave: line (4 k + wave) * 64 + lane
    // of x (k < 6) and of y1 (k < 3).  Why only in the last phase: vmcnt is ONE in-order counter -- a ring wait behind such a touch
    // waits for HBM (round 2 issued them three phases ahead and lost 7 us of chunk phases to exactly that); the last phase's own
    // wait leaves them in flight (vmcnt(NPF)), and the next ring wait is a whole epilogue + row phase away.
    auto prefetch_rows = [&](int t_row0, int t_nrows) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)t_row0 * D), 0, t_nrows * D * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.y1 + (int64_t)t_row0 * D), 0, p.y1 ? t_nrows * D * 2 : 0, 0x00020000);
        int ln;  // (a fresh lane id: the kernel-long one is spilled, and a scratch reload here would wait for the pieces in flight)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const uint32_t vo = (uint32_t)(wave * 8192 + ln * 128);
#pragma unroll
        for (int k = 0; k < 6; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (LDS_AS void*)pfj, 4, vo, k * 32768, 0, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (LDS_AS void*)pfj, 4, vo, k * 32768, 0, 0);
    };
    constexpr int NPF = 9;                   // (what the last phase's ring wait leaves in flight)
    constexpr int NRD = (DBG & 16) ? 2 : 4;  // fragment reads per group

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
    const uint32_t tsbase = (uint32_t)(uintptr_t)(LDS_AS char*)tile_s;

    // ---- prime the ring: units 0 and 1 of the pass ----
    int cons = 0;  // units consumed since kernel start (slot = cons % 3)
    if (tile < p.ntiles) {
        set_issue(0, 0);
        sfor<0, 12>(dma_piece);
        set_issue(1, 1);
        if constexpr (RING2) {
            sfor<0, 12>(dma_piece);  // (both units whole: phase c requests unit c + 2)
        } else {
            sfor<0, 2>(dma_piece);   // (its other ten pieces go out in groups 0..4 of the first phase, as in steady state)
        }
        ipos = 2;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // fragment registers: two sets of 4 (one 4-MFMA group each); group gg of a unit = fragments 4 gg .. 4 gg + 3
    u32x4 wA[2][4];
    auto rd_frag = [&](auto SET_, auto G_, uint32_t sa) __attribute__((always_inline)) {
        constexpr int set = decltype(SET_)::value, gg = decltype(G_)::value;
        const uint32_t a = sa;
        // (asm operands do not trigger the implicit capture in a generic lambda: bind references first)
        u32x4 &d0 = wA[set][0], &d1 = wA[set][1], &d2 = wA[set][2], &d3 = wA[set][3];
        if constexpr (DBG & 8) {
            asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a));
            return;
        }
        DSR128(d0, a, (4 * gg + 0) * 1024);
        DSR128(d1, a, (4 * gg + 1) * 1024);
        if constexpr (DBG & 16) return;  // (ablation: half the fragment reads, the other two MFMAs re-use stale registers)
        DSR128(d2, a, (4 * gg + 2) * 1024);
        DSR128(d3, a, (4 * gg + 3) * 1024);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    f32x4 bq[2][4];  // fc1 bias an A phase starts from: tile U, quad q: b1[off + 32 U + 8 q + 4 h + e], read one phase ahead
    // (offset = a compile-time part, which rides in the instructions' immediates, + a run-time part: hipcc keeps every distinct
    //  b1base + constant in a register of its own across the tile loop, spills it, and reloads it inside a ring phase -- where the
    //  reload's vmcnt(0) waits for the LDS-DMA in flight)
    auto bias_rd = [&](auto OFFC_, int offd) __attribute__((always_inline)) {  // 8 reads, no wait: covered by the next counted wait
        constexpr int oc = decltype(OFFC_)::value * 4;
        int ln;  // (b1[.. + 4 h ..]: the lane half from a fresh lane id -- two instructions -- rather than from a register kept, and spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const uint32_t a = (uint32_t)(uintptr_t)(LDS_AS char*)b1s + ((ln >> 5) << 4) + offd * 4;
        f32x4 &q0 = bq[0][0], &q1 = bq[0][1], &q2 = bq[0][2], &q3 = bq[0][3], &q4 = bq[1][0], &q5 = bq[1][1], &q6 = bq[1][2], &q7 = bq[1][3];
        DSR128(q0, a, oc + 0);
        DSR128(q1, a, oc + 32);
        DSR128(q2, a, oc + 64);
        DSR128(q3, a, oc + 96);
        DSR128(q4, a, oc + 128);
        DSR128(q5, a, oc + 160);
        DSR128(q6, a, oc + 192);
        DSR128(q7, a, oc + 224);
    };

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
        unsigned long long stamp_rt[5] = {0, 0, 0, 0, 0}, stamp_clk[2] = {0, 0};
        PSTAMP(0);
        // next tile: requested now, handed to LDS behind the first row loads (the atomic's round trip is theirs too), read by every
        // wave after the first ring barrier
        int nt_req = 0;
        if (tid == 0) nt_req = atomicAdd(p.counter, 1);
        int tile_next = 0, row0_next = 0, nrows_next = 0;

        u32x4 X[NKS];  // the fc1 B operand: k-step s = 2 c + p, lane half h: columns 32 c + 16 p + 8 (j >> 2) + 4 h + (j & 3)

        f32x16 acc2[NOT];    // the fc2 accumulators, seeded by the row phase with v = x + y1
        f32x16 acc1[2][2];   // [half][tile U]: hidden (reg & 3) + 8 (reg >> 2) + 4 h of tile U for this lane's row
        u32x4 hf[2][2][2];   // [half][tile U][k-step s']: the GELU'd, bf16-packed registers 8 s' .. 8 s' + 7 of acc1[half][U]
        // one 2-element GELU: unit u (0..15) of half GH -> one 32-bit word of the fc2 operand fragments
        auto gelu_unit = [&](auto GH_, auto U_) __attribute__((always_inline)) {
            constexpr int gh = decltype(GH_)::value, u = decltype(U_)::value;
            constexpr int tl = u >> 3, pi = u & 7;
            float v0 = acc1[gh][tl][2 * pi], v1 = acc1[gh][tl][2 * pi + 1];
            if constexpr ((DBG & 2) == 0) {
                v0 = gelu1(v0);
                v1 = gelu1(v1);
            }
            hf[gh][tl][pi >> 2][pi & 3] = pack_bf16x2(v0, v1);
        };

        // ---- one phase: 12 groups of 4 MFMAs on the unit in slot cons % 3 ----
        // KIND/H: fc1 half H (into acc1[H][.], started from the bias in bq) or fc2 half H (operand hf[H][.][.])
        // GH/GSEC: GELU units of half GH, first (0) or second (1) eight, one per group 4..11; GH = -1: none
        // NB/nb: NB > 0: the NEXT phase is an fc1 phase and starts from the bias at b1s offset NB + nb (read with the cross-phase
        //     prefetch).  NB = 0: no bias read.  NB = -1: last phase of the tile, nothing is prefetched (the row phases in between
        //     need the registers)
        auto phase = [&](auto KIND_, auto H_, auto GH_, auto GSEC_, auto NB_, int nb) __attribute__((always_inline)) {
            constexpr int kind = decltype(KIND_)::value, hh = decltype(H_)::value, gh = decltype(GH_)::value;
            constexpr int gsec = decltype(GSEC_)::value, needb = decltype(NB_)::value;
            const uint32_t sa = fbase + (cons % 3) * UNIT;
            const uint32_t sn = fbase + ((cons + 1) % 3) * UNIT;
            if constexpr (RING2) {  // this phase requests unit cons + 2 into the slot unit cons - 1 left at the last barrier
                set_issue(ipos, (cons + 2) % 3);
                ipos = ipos + 1 == upt ? 0 : ipos + 1;
            }
            sfor<0, 12>([&](auto G_) __attribute__((always_inline)) {
                constexpr int gg = decltype(G_)::value, set = gg & 1;
                typedef std::integral_constant<int, set ^ 1> NS;
                // (1) fragment reads one group ahead
                if constexpr (gg == 11) {
                    if constexpr ((DBG & 1) == 0) {
                        // my pieces of the next unit have landed (RING2: all but this phase's own twelve, which are unit cons + 2's;
                        // the last phase's row touches of group 6 are younger still)
                        if constexpr (needb < 0 && PF)
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPF + (RING2 ? 12 : 0)) : "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RING2 ? 12 : 0) : "memory");
                        __builtin_amdgcn_s_barrier();                     // ... everyone's; unit cons-1 is no longer read
                    }
                    if constexpr (!RING2) {
                        set_issue(ipos, (cons + 2) % 3);
                        ipos = ipos + 1 == upt ? 0 : ipos + 1;
                    }
                    if constexpr (needb < 0) {
                        LGKM(0);
                    } else {
                        rd_frag(NS{}, I0{}, sn);
                        if constexpr (needb > 0) {
                            bias_rd(std::integral_constant<int, (needb > 0 ? needb : 0)>{}, nb);
                            LGKM(8 + NRD);
                        } else {
                            LGKM(NRD);
                        }
                    }
                } else {
                    rd_frag(NS{}, std::integral_constant<int, gg + 1>{}, sa);
                    LGKM(NRD);
                }
                // (2) 4 MFMAs, with the vector work that hides under them
                if constexpr (kind == KA) {
                    if constexpr (gg == 0) {
                        // the bias read a phase ago has landed only NOW (the wait above): re-define it here, so that no copy of
                        // it (hipcc moves it to the accumulator file) can be placed before this point
                        f32x4 &q0 = bq[0][0], &q1 = bq[0][1], &q2 = bq[0][2], &q3 = bq[0][3], &q4 = bq[1][0], &q5 = bq[1][1], &q6 = bq[1][2], &q7 = bq[1][3];
                        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(q4), "+v"(q5), "+v"(q6), "+v"(q7));
                        f32x16 t0, t1;  // C operand = bias: register 4 q + e of tile U is hidden 32 U + 8 q + 4 h + e
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                t0[4 * q + e] = bq[0][q][e];
                                t1[4 * q + e] = bq[1][q][e];
                            }
                        mma32<DBG>(t0, wA[set][0], X[0]);
                        mma32<DBG>(t1, wA[set][1], X[0]);
                        mma32<DBG>(t0, wA[set][2], X[1]);
                        mma32<DBG>(t1, wA[set][3], X[1]);
                        acc1[hh][0] = t0;
                        acc1[hh][1] = t1;
                    } else {
                        mma32<DBG>(acc1[hh][0], wA[set][0], X[2 * gg]);
                        mma32<DBG>(acc1[hh][1], wA[set][1], X[2 * gg]);
                        mma32<DBG>(acc1[hh][0], wA[set][2], X[2 * gg + 1]);
                        mma32<DBG>(acc1[hh][1], wA[set][3], X[2 * gg + 1]);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) mma32<DBG>(acc2[gg], wA[set][t], hf[hh][t >> 1][t & 1]);
                }
                if constexpr (RING2) {
                    if constexpr (gg <= 5) {
                        dma_piece(std::integral_constant<int, 2 * gg>{});
                        dma_piece(std::integral_constant<int, 2 * gg + 1>{});
                    }
                } else if constexpr (gg == 11) {
                    dma_piece(std::integral_constant<int, 0>{});
                    dma_piece(std::integral_constant<int, 1>{});
                } else if constexpr (gg <= 4) {
                    dma_piece(std::integral_constant<int, 2 + 2 * gg>{});
                    dma_piece(std::integral_constant<int, 3 + 2 * gg>{});
                }
                if constexpr (needb < 0 && PF && gg == (RING2 ? 6 : 5)) prefetch_rows(row0_next, nrows_next);
                if constexpr (gh >= 0 && gg >= 4) {
                    gelu_unit(std::integral_constant<int, (gh >= 0 ? gh : 0)>{}, std::integral_constant<int, 8 * gsec + gg - 4>{});
                    // (instruction order dealt out by hand, as in mlp16.hip: a 32x32x16 MFMA holds the issue port 8 of its 32 cycles -- five
                    //  or six single-issue instructions fit behind each; hipcc's own order bunches the GELU behind two MFMAs)
                    if constexpr (SGB) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x402, 6, 0);
                        }
                    }
                } else if constexpr (SGB && RING2 && gg <= 5) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            cons += 1;
        };
        typedef std::integral_constant<int, -1> IM1;
        typedef std::integral_constant<int, KA> TA;
        typedef std::integral_constant<int, KB> TB;
        typedef std::integral_constant<int, 64> I64;
        typedef std::integral_constant<int, 128> I128;

        // Where this lane's row lives (row phase and epilogue).  Rows are reached through BUFFER resources over the tile's rows -- a
        // uniform 64-bit base in scalar registers, one 32-bit lane offset, the piece (O, q) in the instruction's scalar offset:
        //   * no 64-bit lane arithmetic: its zero high word is a register hipcc keeps across the whole kernel, spills, and reloads at
        //     the head of the epilogue -- a vmcnt(0) there, behind the row touches above;
        //   * a row past the tile's end gets an offset out of the resource's range: its loads return zeros without traffic and its
        //     stores are dropped -- no predication around 96 stores, no clamped re-reads;
        //   * everything lane-dependent comes from a FRESH lane id: loop-invariant addresses would be hoisted out of the tile loop,
        //     live through the chunk phases, and be spilled there.
        // rb: element offset of the row's columns 4 h .. (row-major forms), fb: of its 16-row fragment (image forms: whole fragments
        // only -- the launcher guarantees M % 16 == 0).
        constexpr uint32_t OOB = 0x80000000u;
        auto row_base = [&](int t_nrows, int& li_, int& h_, uint32_t& rb, uint32_t& fb, bool& live) __attribute__((always_inline)) {
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            li_ = ln & 15;
            h_ = ln >> 5;
            const int m_ = (ln >> 4) & 1;
            const int r = wave * 32 + m_ * 16 + li_;
            live = r < t_nrows;  // (the BYTE offset of a row that is not live becomes OOB: scaled first, or it would wrap back into range)
            rb = (uint32_t)(r * D + 4 * h_);
            fb = (uint32_t)((wave * 32 + m_ * 16) * D);
        };
        // both lanes of a row (l, l ^ 32) get lo + hi, summed in that order (no LDS crossbar, no lane-id register)
        auto row_sum = [&](float v) __attribute__((always_inline)) -> float {
#pragma clang fp contract(off)
            const uint32_t u = __builtin_bit_cast(uint32_t, v);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            return __builtin_bit_cast(float, (uint32_t)sw[0]) + __builtin_bit_cast(float, (uint32_t)sw[1]);
        };

        // ---- row phase: v = x + y1 in the ACCUMULATOR layout.  Lane (h, m, li) holds row 32 w + 16 m + li and, per output tile O and
        // quad q, columns 32 O + 8 q + 4 h + (0..3): the 16-byte pieces of the fp32 row (or image), the 8-byte pieces of the bf16
        // one.  The two h-lanes of a row hold all of it, so LayerNorm-2 takes ONE cross-lane step; the normalised values packed to
        // bf16 ARE the fc1 operand (the fc1 weight image lists k in the accumulator's column order: mlp32_pack_kernel); and v seeds
        // the fc2 accumulators, so the epilogue never re-reads x and y1 (round 3 did: 1.21 of the 3.99 GB per 8-region launch).
        // All 96 loads of a lane (36 KiB per wave) are in flight together: one memory latency per tile instead of two.
        {
#pragma clang fp contract(off)
            uint32_t xo, yo;  // byte offsets of this lane's pieces inside the tile's x / y1 rows
            uint32_t g2base;  // LN-2 gamma in accumulator column order: gam[32 O + 8 q + 4 h ..] at + (32 O + 8 q) * 4 (beta: + D * 4)
            {
                uint32_t rb, fb;
                int li_, h_;
                bool live;
                row_base(nrows, li_, h_, rb, fb, live);
                xo = live ? (XIN ? fb + (uint32_t)(256 * h_ + 4 * li_) : rb) * 4 : OOB;
                yo = live ? (IMG ? fb + (uint32_t)(8 * li_ + 4 * h_) : rb) * 2 : OOB;
                g2base = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 16 * h_;
            }
            constexpr int xlo_ = XIN ? 512 : 32, xlq_ = XIN ? 64 : 8, yo_ = IMG ? 512 : 32, yq_ = IMG ? 128 : 8;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)row0 * D), 0, nrows * D * 4, 0x00020000);
            // (no y1: an empty range -- every piece reads as zero)
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.y1 + (int64_t)row0 * D), 0, p.y1 ? nrows * D * 2 : 0, 0x00020000);
            f32x4 xv[NOT][4];
            u32x2 yv[NOT][4];
#pragma unroll
            for (int O = 0; O < NOT; ++O)
#pragma unroll
                for (int q = 0; q < 4; ++q) xv[O][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo, (xlo_ * O + xlq_ * q) * 4, 0));
#pragma unroll
            for (int O = 0; O < NOT; ++O)
#pragma unroll
                for (int q = 0; q < 4; ++q) yv[O][q] = __builtin_amdgcn_raw_buffer_load_b64(ry, yo, (yo_ * O + yq_ * q) * 2, 0);
            if (tid == 0) {
                asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt_req) : "memory");
                if (nt_req == last_fetch) *p.counter = 0;
            }
            // (packed fp32 arithmetic: no MFMA runs beside the row phases, and it halves their vector instructions)
            f32x2 rs2 = {0.f, 0.f};
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int O = decltype(O_)::value;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x4 y = __builtin_bit_cast(bf16x4, yv[O][q]);
                    f32x2 a = {xv[O][q][0], xv[O][q][1]}, b = {xv[O][q][2], xv[O][q][3]};
                    a = a + f32x2{(float)y[0], (float)y[1]};
                    b = b + f32x2{(float)y[2], (float)y[3]};
                    rs2 = rs2 + a;
                    rs2 = rs2 + b;
                    xv[O][q] = f32x4{a[0], a[1], b[0], b[1]};
                }
            });
            const float rs = row_sum(rs2[0] + rs2[1]);
            const float mean = rs * (1.0f / D);
            const f32x2 mean2 = {mean, mean};
            f32x2 qs2 = {0.f, 0.f};
#pragma unroll
            for (int O = 0; O < NOT; ++O)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 a = f32x2{xv[O][q][0], xv[O][q][1]} - mean2, b = f32x2{xv[O][q][2], xv[O][q][3]} - mean2;
                    qs2 = __builtin_elementwise_fma(a, a, qs2);
                    qs2 = __builtin_elementwise_fma(b, b, qs2);
                }
            const float qs = row_sum(qs2[0] + qs2[1]);
            const float rstd = 1.0f / sqrtf(qs * (1.0f / D) + p.ln_eps);
            const f32x2 rstd2 = {rstd, rstd};
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int O = decltype(O_)::value;
                f32x4 gq[4], bqv[4];
                const uint32_t ga = g2base;
                f32x4 &g0 = gq[0], &g1 = gq[1], &g2 = gq[2], &g3 = gq[3], &b0 = bqv[0], &b1_ = bqv[1], &b2_ = bqv[2], &b3 = bqv[3];
                DSR128X4_WAIT(g0, g1, g2, g3, ga, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
                DSR128X4_WAIT(b0, b1_, b2_, b3, ga, D * 4 + O * 128, D * 4 + O * 128 + 32, D * 4 + O * 128 + 64, D * 4 + O * 128 + 96);
                uint32_t w[8];
                f32x16 t;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 va = {xv[O][q][0], xv[O][q][1]}, vb = {xv[O][q][2], xv[O][q][3]};
                    const f32x2 ya = __builtin_elementwise_fma((va - mean2) * rstd2, f32x2{gq[q][0], gq[q][1]}, f32x2{bqv[q][0], bqv[q][1]});
                    const f32x2 yb = __builtin_elementwise_fma((vb - mean2) * rstd2, f32x2{gq[q][2], gq[q][3]}, f32x2{bqv[q][2], bqv[q][3]});
                    w[2 * q] = pack_bf16x2(ya[0], ya[1]);
                    w[2 * q + 1] = pack_bf16x2(yb[0], yb[1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[4 * q + e] = xv[O][q][e];
                }
                X[2 * O] = u32x4{w[0], w[1], w[2], w[3]};      // k-step 2 O: columns 8 q + 4 h + e of the 32, q = 0, 1
                X[2 * O + 1] = u32x4{w[4], w[5], w[6], w[7]};  // k-step 2 O + 1: q = 2, 3
                asm volatile("" : "+a"(t));  // the seed goes to the accumulator file at once
                acc2[O] = t;
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        PSTAMP(2);
        PSTAMP_CLK(0);

#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) hf[a][b][c] = u32x4{0u, 0u, 0u, 0u};
        // first fragments and bias of the pass (asm reads land asynchronously: nothing but the first phase may sit
        // between them and their counted wait -- in particular not the row phases, where the compiler moves registers)
        rd_frag(I0{}, I0{}, fbase + (cons % 3) * UNIT);
        bias_rd(I0{}, 0);
        // chunk 0 (peeled: no runtime branches around phases inside the steady-state loop).  Its half-0 GELUs have
        // only A1(0) to hide in: the second eight run bare.
        phase(TA{}, I0{}, IM1{}, I0{}, I64{}, 0);
        phase(TA{}, I1{}, I0{}, I0{}, I0{}, 0);
        sfor<8, 16>([&](auto U_) __attribute__((always_inline)) { gelu_unit(I0{}, U_); });
        __builtin_amdgcn_sched_barrier(0);
        {
            // the tile after this one (handed over before the first ring barrier): which rows the prefetch below and the next pass
            // of the loop work on.  The wait also covers the fragments the last phase requested ahead: a few hundred cycles, once a tile
            int nt;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nt) : "v"(tsbase + 4 * ((seq + 1) & 1)) : "memory");
            tile_next = __builtin_amdgcn_readfirstlane(nt);
            if (tile_next < p.full_tiles) {
                row0_next = tile_next * TMR;
                nrows_next = TMR;
            } else {
                row0_next = p.full_tiles * TMR + (tile_next - p.full_tiles) * 16;
                nrows_next = 16;
            }
            nrows_next = (p.M - row0_next) < nrows_next ? (p.M - row0_next) : nrows_next;
            nrows_next = (tile_next < p.ntiles && nrows_next > 0) ? nrows_next : 0;  // (no next tile: an empty range, every load dropped)
            row0_next = nrows_next > 0 ? row0_next : 0;
        }
        phase(TB{}, I0{}, I1{}, I0{}, I128{}, 0);
        for (int c = 1; c < nchunk - 1; ++c) {
            phase(TA{}, I0{}, I1{}, I1{}, I0{}, 0);             // A0(c)   + second eight GELUs of half 1 of chunk c-1
            phase(TB{}, I1{}, I0{}, I0{}, I64{}, c * 128);    // B1(c-1) + first eight of half 0 of chunk c
            phase(TA{}, I1{}, I0{}, I1{}, I0{}, 0);             // A1(c)   + second eight of half 0
            phase(TB{}, I0{}, I1{}, I0{}, I128{}, c * 128);   // B0(c) + first eight of half 1
        }
        {   // the last chunk (peeled): its first two phases also request the rows of the epilogue and of the next tile's row phase
            const int c = nchunk - 1;
            phase(TA{}, I0{}, I1{}, I1{}, I0{}, 0);
            phase(TB{}, I1{}, I0{}, I0{}, I64{}, c * 128);
            phase(TA{}, I1{}, I0{}, I1{}, I0{}, 0);
            phase(TB{}, I0{}, I1{}, I0{}, I128{}, -128);  // (a bias nobody uses: the wait counts stay those of the loop body)
        }
        LGKM(0);  // (the last B0 read a bias nobody uses: let it land before its registers are re-used ...
        {         //  ... and keep those registers allocated up to here: a fake use AFTER the wait)
            f32x4 &q0 = bq[0][0], &q1 = bq[0][1], &q2 = bq[0][2], &q3 = bq[0][3], &q4 = bq[1][0], &q5 = bq[1][1], &q6 = bq[1][2], &q7 = bq[1][3];
            asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7));
        }
        // tail: second eight of the last half 1, then B1(last); its prefetch is the next tile's A0(0)
        sfor<8, 16>([&](auto U_) __attribute__((always_inline)) { gelu_unit(I1{}, U_); });
        __builtin_amdgcn_sched_barrier(0);
        phase(TB{}, I1{}, IM1{}, I0{}, IM1{}, 0);
        PSTAMP(3);
        PSTAMP_CLK(1);

        // ---- epilogue: x <- acc2 + b2 (acc2 started from v = x + y1: nothing is re-read).  This workgroup owns its rows: in place.
        //      Lane (h, m, li) holds row 32 w + 16 m + li; acc2[O][4 q + e] is output column 32 O + 8 q + 4 h + e.
        //      Images: chunk-of-8 index 4 O + q = g' + 4 c' with g' = q, c' = O, half h, image lane 16 q + li.
        //      (row-major in, image out -- the first block of a forward -- converts in place: a wave's 32 rows are the bytes of its two
        //       fragments, and every old value was loaded in the row phase)
        {
#pragma clang fp contract(off)
            uint32_t rb, fb;
            int li_, h_;
            bool live;
            row_base(nrows, li_, h_, rb, fb, live);
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)row0 * D), 0, nrows * D * 4, 0x00020000);
            const uint32_t b2base = (uint32_t)(uintptr_t)(LDS_AS char*)b2s + 16 * h_;   // b2[32 O + 8 q + 4 h ..]: + (32 O + 8 q) * 4
            const uint32_t g1base = (uint32_t)(uintptr_t)(LDS_AS char*)gam1 + 16 * h_;  // next LN-1 gamma (beta: + D * 4)
            // float / element offsets of piece (O, q): row-major rb + 32 O + 8 q; fp32 image fb + 512 O + 256 h + 64 q + 4 li;
            // bf16 image fb + 512 O + 128 q + 8 li + 4 h
            const uint32_t xso = live ? (IMG ? fb + (uint32_t)(256 * h_ + 4 * li_) : rb) * 4 : OOB;
            constexpr int xso_ = IMG ? 512 : 32, xsq_ = IMG ? 64 : 8, yo_ = IMG ? 512 : 32, yq_ = IMG ? 128 : 8;
            f32x2 rs2 = {0.f, 0.f};
            sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int O = decltype(O_)::value;
                f32x4 bb[4];
                const uint32_t ba = b2base;
                f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
                f32x16 t = acc2[O];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 a = f32x2{t[4 * q], t[4 * q + 1]} + f32x2{bb[q][0], bb[q][1]};
                    const f32x2 b = f32x2{t[4 * q + 2], t[4 * q + 3]} + f32x2{bb[q][2], bb[q][3]};
                    rs2 = rs2 + a;
                    rs2 = rs2 + b;
                    const f32x4 v = {a[0], a[1], b[0], b[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rx, xso, (xso_ * O + xsq_ * q) * 4, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[4 * q + e] = v[e];
                }
                acc2[O] = t;
                __builtin_amdgcn_sched_barrier(0);
            });
            if (p.xn_out) {
                // LayerNorm-1 of the next block on the finished row (the two h-lanes of a row hold all of it), as bf16
                const float rs = row_sum(rs2[0] + rs2[1]);
                const float mean = rs * (1.0f / D);
                const f32x2 mean2 = {mean, mean};
                f32x2 qs2 = {0.f, 0.f};
#pragma unroll
                for (int O = 0; O < NOT; ++O)
#pragma unroll
                    for (int e = 0; e < 16; e += 2) {
                        const f32x2 a = f32x2{acc2[O][e], acc2[O][e + 1]} - mean2;
                        qs2 = __builtin_elementwise_fma(a, a, qs2);
                    }
                const float qs = row_sum(qs2[0] + qs2[1]);
                const float rstd = 1.0f / sqrtf(qs * (1.0f / D) + p.ln_eps);
                const f32x2 rstd2 = {rstd, rstd};
                const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)p.xn_out + (int64_t)row0 * D), 0, nrows * D * 2, 0x00020000);
                const uint32_t nso = live ? (IMG ? fb + (uint32_t)(8 * li_ + 4 * h_) : rb) * 2 : OOB;
                sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int O = decltype(O_)::value;
                    f32x4 gq[4], bqv[4];
                    const uint32_t ga = g1base;
                    f32x4 &g0 = gq[0], &g1 = gq[1], &g2 = gq[2], &g3 = gq[3], &b0 = bqv[0], &b1_ = bqv[1], &b2_ = bqv[2], &b3 = bqv[3];
                    DSR128X4_WAIT(g0, g1, g2, g3, ga, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
                    DSR128X4_WAIT(b0, b1_, b2_, b3, ga, D * 4 + O * 128, D * 4 + O * 128 + 32, D * 4 + O * 128 + 64, D * 4 + O * 128 + 96);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 va = {acc2[O][4 * q], acc2[O][4 * q + 1]}, vb = {acc2[O][4 * q + 2], acc2[O][4 * q + 3]};
                        const f32x2 ya = __builtin_elementwise_fma((va - mean2) * rstd2, f32x2{gq[q][0], gq[q][1]}, f32x2{bqv[q][0], bqv[q][1]});
                        const f32x2 yb = __builtin_elementwise_fma((vb - mean2) * rstd2, f32x2{gq[q][2], gq[q][3]}, f32x2{bqv[q][2], bqv[q][3]});
                        u32x2 o2;
                        o2[0] = pack_bf16x2(ya[0], ya[1]);
                        o2[1] = pack_bf16x2(yb[0], yb[1]);
                        __builtin_amdgcn_raw_buffer_store_b64(o2, rn, nso, (yo_ * O + yq_ * q) * 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
        PSTAMP(4);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) {
            unsigned long long* o = p.stamps + (size_t)blockIdx.x * 16;
            o[0] = stamp_rt[0];
            o[2] = stamp_rt[2];
            o[3] = stamp_rt[3];
            o[4] = stamp_rt[4];
            o[8] = stamp_clk[0];
            o[9] = stamp_clk[1];
        }
        tile = tile_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: pieces of a pass that never runs)
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

bool hipt_mlp32_supported(int dtype, int D_, int hidden) {
    return dtype == HIPT_BF16 && D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536;
}

int hipt_mlp32_pack_launch(const void* w1, const void* w2, int D_, int hidden, void* packed, hipStream_t st) {
    if (!(D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536)) {
        hipt_set_error("mlp32 pack: unsupported D=%d hidden=%d", D_, hidden);
        return HIPT_E_UNSUPPORTED;
    }
    const int64_t chunks = (int64_t)(hidden / 128) * 4 * (UNIT / 16);
    hipLaunchKernelGGL(mlp32_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w1, (const bf16_t*)w2, hidden, (u32x4*)packed);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

template <int DBG>
int hipt_mlp32_launch_dbg(const MlpParams& p_in, hipStream_t st) {
    MlpParams p = p_in;
    const int lds = 3 * UNIT + (3 * D + p.hidden) * 4 + 16 + 2 * D * 4 + 256;
    if (!p.wpk || p.wpk_fmt != 1 || (p.img & 2 && !(p.img & 1)) || (p.img && p.M % 16 != 0) || p.fold) {
        hipt_set_error("mlp32: needs its packed weights; activation images need M %% 16 == 0 and img in {0, 1, 3}; no proj folding (img=%d, M=%d, fold=%d)", p.img,
                       p.M, p.fold);
        return HIPT_E_BADARG;
    }
    auto k = p.img == 3 ? mlp32_kernel<true, true, DBG> : p.img == 1 ? mlp32_kernel<true, false, DBG> : mlp32_kernel<false, false, DBG>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)mlp32_kernel<true, true, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp32_kernel<true, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp32_kernel<false, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(mlp32) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("mlp32: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    const int ncu = once.ncu[dev];
    // Whole rounds of #CU workgroups take 128 rows each.  A last partial round less than an eighth full: in a short launch (up to 4
    // rounds: one or two regions per call) it is cut into 16-row tiles on 8x the CUs -- same pass over the weights, a fraction of
    // the row phases; in a long one the leftover tiles stay whole on their few CUs, which leaves the others to the next kernel
    // of another stream (HIPT_4K spreads its regions over streams: +1.4 % regions/s at 8 regions per stream).
    const int tiles = (p.M + TMR - 1) / TMR;
    const int rem = tiles % ncu;
    // (and a launch of at most an eighth of a round -- the [CLS] rows of the pruned last block: 2 tiles at one region per call, 16 at
    //  eight -- is all 16-row tiles: 8x the CUs, each with a pass over the weights and a fraction of the row phases)
    const int tail_tiles = tiles <= ncu / 8 ? tiles : ((tiles > ncu && tiles <= 4 * ncu + ncu / 8 && rem > 0 && rem <= ncu / 8) ? rem : 0);
    p.full_tiles = tiles - tail_tiles;
    const int tail_rows = p.M - p.full_tiles * TMR;
    p.ntiles = p.full_tiles + (tail_rows > 0 ? (tail_rows + 15) / 16 : 0);
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    p.stagger = 0;
    if (!p.counter_zeroed && hipMemsetAsync(p.counter, 0, sizeof(int), st) != hipSuccess) {
        hipt_set_error("mlp32: hipMemsetAsync(counter) failed");
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
        double pro = 0, chunks = 0, epi = 0, ghz = 0;
        for (int b = 0; b < grid; ++b) {
            pro += (double)(h[b * 16 + 2] - h[b * 16 + 0]) * 0.01 / grid;
            chunks += (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.01 / grid;
            epi += (double)(h[b * 16 + 4] - h[b * 16 + 3]) * 0.01 / grid;
            ghz += (double)(h[b * 16 + 9] - h[b * 16 + 8]) / (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.1 / grid;
        }
        fprintf(stderr, "[mlp32 dbg=%d hidden=%d grid=%d tiles=%d(+%d)] total %.1f us | tile %d of each workgroup: rows+LN %.1f, chunks %.1f (%.2f GHz), epilogue %.1f\n",
                DBG, p.hidden, grid, p.full_tiles, p.ntiles - p.full_tiles, (double)(t4 - t0) * 0.01, PSTAMP_SEQ, pro, chunks, ghz, epi);
    }
#endif
    return HIPT_OK;
}

int hipt_mlp32_launch(const MlpParams& p, hipStream_t st) { return hipt_mlp32_launch_dbg<0>(p, st); }
