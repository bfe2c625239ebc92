// FUSED MLP SUB-BLOCK for D = 384 (ViT-256) on 16x16x32 MFMAs:   x <- x + y1 + fc2( GELU( fc1( LN2(x + y1) ) ) )
//   (Block.forward second half, HIPT_4K/vision_transformer.py:151 with Mlp.forward :98-104.)
//
// The same kernel as mlp32.hip -- one 4-wave workgroup owns 128 rows, every wave its 32 rows end to end in registers, weights
// through the 3 x 48 KiB LDS-DMA ring, phases A0(c) B1(c-1) A1(c) B0(c), row phase in the accumulator layout seeding the fc2
// accumulators, epilogue without re-reads -- on the OTHER bf16 MFMA shape.  Why (round 4, tools/mfma_shape_probe.hip, random operands,
// steady-state clock): a bare v_mfma_f32_16x16x32_bf16 loop holds 2.35 GHz where the 32x32x16 loop is throttled to 1.9 GHz --
// 2 220 against 1 926 TFLOP/s, 1 968 against 1 538 with one ds_read_b128 per 32 matrix-pipe cycles.  These kernels run at the
// chip's power cap (every re-scheduling of mlp32.hip's work lands on the same time: DESIGN.md), so energy per FLOP is what counts.
//   * a wave's 32 rows are TWO B operands (row fragments m = 0 / 1 of 16 rows); every 1 KiB weight fragment (16 units x 32 k) is
//     read from LDS once and feeds two MFMAs of 16 cycles: the LDS bytes per FLOP of mlp32.hip.
//   * lane (g, i) = (lane >> 4, lane & 15).  B operand: row i of the fragment, k-slot group g.  Accumulator tile (16 units x 16
//     rows): lane holds units 4 g + (0..3) of row i -- 16-byte pieces of an fp32 row at column 16 T + 4 g.
//   * k orders follow the accumulator's columns, so that an accumulator tile pair IS an operand: k-step s of lane group g carries
//     columns 32 s + 16 (j >> 2) + 4 g + (j & 3), j = 0..7 -- for the fc1 operand built by the row phase from LayerNorm-2'd
//     accumulator-layout values, and for the fc2 operand packed from GELU'd fc1 accumulator tiles 2 t', 2 t' + 1.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "mlp_common.h"
#include "pipe_common.h"

// Cache policy of the row traffic (the builtin's aux operand: 2 = nt).  A tile's rows are read once and written once per launch (2.4 GB per 2 048 patches)
// while every tile re-reads the 2.4 MB weight image from L2: the rows are marked non-temporal so that they do not push the image out.  Round 5, same box,
// two runs each: 1 430 / 1 428 us per launch without, 1 414 / 1 418 with (and the attention kernel behind it 685 / 681 -> 675 / 673: it finds its own
// weight image in L2 more often); 325.4 / 325.8 -> 326.6 / 327.6 regions/s; HBM bytes per launch by PMC 2.78 -> 2.47 GB (2.43 algorithmic).  IMAGE
// layouts only (whole 1 KiB per instruction): row-major pieces are 16 bytes of a 128-byte line that the lane's next pieces need again -- the block-1
// form (row-major x from the patch embedding) got 12 % SLOWER with the hint on its x loads (1 470 -> 1 648 us).
constexpr int AUX_ROWS = 2;

namespace {

constexpr int D = 384, NKS = 12, NT16 = 24, TMR = 128;   // NKS: 32-deep k-steps of fc1; NT16: 16-wide output tiles
constexpr int UNIT = 48 * 1024;                                    // ring unit = one phase = 48 fragments of 1 KiB

template <int DBG = 0>
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b) {
    if constexpr (DBG & 4) {
        asm volatile("" : "+v"(acc) : "v"(a), "v"(b));
        return;
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
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

#define DSR128A(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "n"(off))

enum { KA = 0, KB = 1, KP = 2 };  // phase kind: fc1 half / fc2 half / proj unit (FOLD: H = unit 0..5, operand = the attention rows in X)

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

// The packed image: unit after unit in pass order, each 48 fragments x 1 KiB, lane-major (lane l = 16 g + i: 16 bytes at l * 16).
//   fc1 unit (chunk c, half hh: hidden Hb = 128 c + 64 hh): fragment 4 s + U (k-step s 0..11, hidden tile U 0..3): element j =
//       W1[Hb + 16 U + i][32 s + 16 (j >> 2) + 4 g + (j & 3)]
//   fc2 unit: fragment 4 gg + f (group gg 0..11; f = 2 (O' & 1) + t': output tile O' = 2 gg + (f >> 1), k-step t' 0/1): element j =
//       W2[16 O' + i][Hb + 32 t' + 16 (j >> 2) + 4 g + (j & 3)]
//   proj units (format 3 only: six units IN FRONT of the pass, wp = the proj matrix [D, D]): unit pu, fragment 4 gg + f as an fc2 unit (output tile
//       O' = 2 gg + (f >> 1), k-step ks = 2 pu + (f & 1)): element j = Wproj[16 O' + i][32 ks + 8 g + j] -- 8 CONSECUTIVE columns per lane group, the
//       chunk order in which the attention output (row-major or image) is loaded as the other operand
__global__ void mlp16_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2, const bf16_t* __restrict__ wp, int hidden,
                                  u32x4* __restrict__ out) {
    const int nchunk = hidden / 128, upt = 4 * nchunk, npu = wp ? 6 : 0;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte lane chunk
    if (i >= (int64_t)(upt + npu) * (UNIT / 16)) return;
    const int pos = (int)(i / (UNIT / 16)) - npu, o = (int)(i % (UNIT / 16)), frag = o >> 6, lane = o & 63, r = lane & 15, g = lane >> 4;
    if (pos < 0) {
        const int pu = pos + npu, gg = frag >> 2, f = frag & 3, O = 2 * gg + (f >> 1), ks = 2 * pu + (f & 1);
        out[i] = *(const u32x4*)(wp + (int64_t)(16 * O + r) * D + 32 * ks + 8 * g);
        return;
    }
    bool is_a;
    int c, hh;
    unit_of(pos, nchunk, is_a, c, hh);
    const int Hb = 128 * c + 64 * hh;
    const bf16_t* row;
    if (is_a) {
        const int ks = frag >> 2, U = frag & 3;
        row = w1 + (int64_t)(Hb + 16 * U + r) * D + 32 * ks + 4 * g;
    } else {
        const int gg = frag >> 2, f = frag & 3, O = 2 * gg + (f >> 1), t = f & 1;
        row = w2 + (int64_t)(16 * O + r) * hidden + Hb + 32 * t + 4 * g;
    }
    // The image carries W1 / 8 and 8 W2 (exact: powers of two; b1 / 8 is made where the kernel stages the bias): fc1 then yields h / 8, whose square the
    // GELU clamps at 1 with the multiply's own clamp bit instead of a v_min at 64 (mlp_common.h, gelu1s), its output g / 8 meets 8 W2 -- every
    // product, sum and rounding of the unscaled computation, bit for bit, one vector instruction per hidden element less.
    const float sc = is_a ? 0.125f : 8.0f;
    u32x4 ov;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const bf16_t* q = row + 16 * h;
        ov[2 * h] = pack_bf16x2((float)q[0] * sc, (float)q[1] * sc);
        ov[2 * h + 1] = pack_bf16x2((float)q[2] * sc, (float)q[3] * sc);
    }
    out[i] = ov;
}

// IMG / XIN: fragment-blocked activation images (kernels.h, "activation images") -- IMG: y1 is read and x / xn_out are written as
// images; XIN: x is read as an image.  Row-major otherwise.  Weights always come from the packed image p.wpk (format 1).
// DBG (tools/mlp_probe.hip only): 1 = no weight DMA / ring syncs, 2 = GELU replaced by a plain pack, 4 = no MFMAs, 8 = no LDS
// fragment reads.
// IN PLACE: p.xn_out may be the buffer p.y1 points to (hipt_vit_mlp_unit documents it for callers; run_blocks itself never aliases them since
// round 6).  What makes that safe, and what any change to this kernel must keep: a workgroup loads ALL rows of its tile (x and y1 / the
// attention rows) in the row phase, before the first store of its epilogue; tiles own disjoint rows; nothing is prefetched from another tile's
// y1 rows (the L2 touches of DBG 64 only read).  A next-tile prefetch of y1 into registers, or an epilogue that stores before the row phase
// has consumed its loads, would need distinct buffers.
// FOLD (round 5; image format 3): the attention block's output projection runs at the head of the tile -- p.y1 is then the ATTENTION OUTPUT (before
// proj; row-major or image like y1), six more ring units in front of the pass hold the proj matrix, the fc2 accumulators start from x + b_proj, take
// att Wproj^T through six KB-shaped phases (the attention rows sit in X's registers: same shape), and v = x + proj(att) + b is what LayerNorm-2 reads
// back from them: the proj launch, its 0.8 GB of HBM traffic per 2 048 patches and the bf16 rounding of y1 are gone.
template <bool IMG = false, bool XIN = false, int DBG = 0, bool FOLD = false>
__global__ __launch_bounds__(256, 1) void mlp16_kernel(const MlpParams p) {
    // RING2 (round 4): TWO weight units in flight.  Rounds 1-3 requested unit c + 1 in groups 0..4 of phase c and waited for it at the
    // end of the same phase: the last pieces had 7 groups (~0.8 us) to come from L2 -- under the load of 256 CUs streaming the image
    // that is the L2 -> LDS latency itself, every phase ended in that wait, and whatever the waves did in between (MFMA shape, GELU
    // form, LDS reads, row prefetches) landed on the same time; the ring's third slot held a dead unit all the while.  Now phase c
    // requests unit c + 2 into that slot and its closing wait is vmcnt(12): a unit has a whole phase more to land.
    // (DBG 128, tools/mlp_probe.hip: the old protocol)
    constexpr bool RING2 = (DBG & 128) == 0;
    constexpr bool SGB = (DBG & 256) == 0;  // (DBG 256: hipcc's own instruction order inside the GELU groups)
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
    float* bps = pfj + 64;                 // [D] proj bias (FOLD)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nchunk = p.hidden / 128;
    const int upt = 4 * nchunk + (FOLD ? 6 : 0);  // ring units (phases) per tile pass

    // ---- weight DMA: unit pos of the image = 48 pieces of 1 KiB, byte for byte what its ring slot holds; wave w issues pieces
    // 12 w .. 12 w + 11.  An LDS-DMA instruction takes its LDS base from M0, and it is WRITING M0 that makes a piece expensive
    // (tools/issue_mix_probe.hip: +36 cycles per piece with a new M0, +2 with the same M0 and the piece selected by the instruction's
    // immediate offset, which is added to the LDS and to the global address alike): four consecutive pieces share one M0.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 2 * p.hidden * D * 2 + (FOLD ? D * D * 2 : 0), 0x00020000);
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
        if constexpr (FOLD) bps[i] = p.bproj[i];
        if (p.xn_out) {
            gam1[i] = p.ln_next_w[i];
            gam1[D + i] = p.ln_next_b[i];
        }
    }
    for (int i = tid; i < p.hidden; i += 256) b1s[i] = p.b1[i] * 0.125f;  // (the image holds W1 / 8: mlp16_pack_kernel)
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

    f32x4 bq[4];  // fc1 bias an A phase starts from: tile U: b1[off + 16 U + 4 g + e], read one phase ahead
    // (offset = a compile-time part, which rides in the instructions' immediates, + a run-time part: hipcc keeps every distinct
    //  b1base + constant in a register of its own across the tile loop, spills it, and reloads it inside a ring phase -- where the
    //  reload's vmcnt(0) waits for the LDS-DMA in flight)
    auto bias_rd = [&](auto OFFC_, int offd) __attribute__((always_inline)) {  // 4 reads, no wait: covered by the next counted wait
        constexpr int oc = decltype(OFFC_)::value * 4;
        int ln;  // (b1[.. + 4 g ..]: the lane group from a fresh lane id -- two instructions -- rather than from a register kept, and spilled)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const uint32_t a = (uint32_t)(uintptr_t)(LDS_AS char*)b1s + ((ln >> 4) << 4) + offd * 4;
        // (straight into the accumulator file: the bias is the C operand of the phase's first MFMAs, and every MFMA destination of this 512-register
        //  kernel lives there -- read into arch registers it cost 16 v_accvgpr_write per fc1 phase)
        f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
        DSR128A(q0, a, oc + 0);
        DSR128A(q1, a, oc + 64);
        DSR128A(q2, a, oc + 128);
        DSR128A(q3, a, oc + 192);
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

        u32x4 X[2][NKS];  // the fc1 B operands: [row fragment m][k-step s]: lane group g: columns 32 s + 16 (j >> 2) + 4 g + (j & 3)

        f32x4 acc2[NT16][2];   // the fc2 accumulators [output tile T][m], seeded by the row phase with v = x + y1
        f32x4 acc1[2][4][2];   // [half][hidden tile U][m]: hidden 16 U + 4 g + (0..3) of the half for this lane's row
        u32x4 hf[2][2][2];     // [half][k-step t'][m]: the GELU'd, bf16-packed accumulator tiles 2 t', 2 t' + 1
        // one 2-element GELU: unit u (0..15) of half GH -> one 32-bit word of the fc2 operand fragments
        auto gelu_unit = [&](auto GH_, auto U_) __attribute__((always_inline)) {
            constexpr int gh = decltype(GH_)::value, u = decltype(U_)::value;
            constexpr int tl = u >> 2, mm = (u >> 1) & 1, pr = u & 1;
            float v0 = acc1[gh][tl][mm][2 * pr], v1 = acc1[gh][tl][mm][2 * pr + 1];
            if constexpr ((DBG & 2) == 0) {
                v0 = gelu1s(v0);
                v1 = gelu1s(v1);
            }
            hf[gh][tl >> 1][mm][(tl & 1) * 2 + pr] = pack_bf16x2(v0, v1);
        };

        // ---- one phase: 12 groups of 4 fragments = 8 MFMAs on the unit in slot cons % 3 ----
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
                            LGKM(4 + NRD);
                        } else {
                            LGKM(NRD);
                        }
                    }
                } else {
                    rd_frag(NS{}, std::integral_constant<int, gg + 1>{}, sa);
                    LGKM(NRD);
                }
                // (2) 8 MFMAs (4 fragments x 2 row fragments), with the vector work that hides under them
                if constexpr (kind == KA) {  // group gg = k-step gg: hidden tiles U = 0..3
                    if constexpr (gg == 0) {
                        // the bias read a phase ago has landed only NOW (the wait above): re-define it here, so that no copy of
                        // it (hipcc moves it to the accumulator file) can be placed before this point
                        f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
                        asm volatile("" : "+a"(q0), "+a"(q1), "+a"(q2), "+a"(q3));
#pragma unroll
                        for (int U = 0; U < 4; ++U)
#pragma unroll
                            for (int mm = 0; mm < 2; ++mm) {
                                f32x4 t = bq[U];  // C operand = bias: register e of tile U is hidden 16 U + 4 g + e
                                mma16<DBG>(t, wA[set][U], X[mm][0]);
                                acc1[hh][U][mm] = t;
                            }
                    } else {
#pragma unroll
                        for (int U = 0; U < 4; ++U)
#pragma unroll
                            for (int mm = 0; mm < 2; ++mm) mma16<DBG>(acc1[hh][U][mm], wA[set][U], X[mm][gg]);
                    }
                } else {  // group gg = output tiles 2 gg, 2 gg + 1 x k-steps t' = 0, 1
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm) {
                            if constexpr (kind == KP) mma16<DBG>(acc2[2 * gg + (f >> 1)][mm], wA[set][f], X[mm][2 * hh + (f & 1)]);
                            else mma16<DBG>(acc2[2 * gg + (f >> 1)][mm], wA[set][f], hf[hh & 1][f & 1][mm]);
                        }
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
                    // The wave issues in order: a 16x16x32 MFMA holds the issue port 8 of its 16 cycles, so two or three single-issue
                    // instructions fit behind each one for free -- but hipcc's own order for this group is MFMA, MFMA, twelve GELU
                    // instructions in a row (the matrix pipe idle under them), ..., three MFMAs back to back (the wave stalled on the
                    // pipe): the counters show it (38 % of the wave cycles in issue stalls beside 44 % matrix-pipe occupancy).
                    // Dealt out instead: one MFMA, three vector instructions (transcendentals included), eight times.
                    if constexpr (SGB) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x402, 3, 0);
                        }
                    }
                }
                if constexpr (SGB && !(gh >= 0 && gg >= 4) && RING2 && gg <= 5) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            cons += 1;
        };
        typedef std::integral_constant<int, -1> IM1;
        typedef std::integral_constant<int, KA> TA;
        typedef std::integral_constant<int, KB> TB;
        typedef std::integral_constant<int, KP> TP;
        typedef std::integral_constant<int, 64> I64;
        typedef std::integral_constant<int, 128> I128;

        // Where this lane's rows live (row phase and epilogue).  Lane (g, i) holds row i of BOTH row fragments of its wave (rows
        // 32 w + 16 m + i, m = 0 / 1) and, per output tile T, columns 16 T + 4 g + (0..3): a 16-byte piece of the fp32 row (or image),
        // an 8-byte piece of the bf16 one.  Rows are reached through BUFFER resources over the tile's rows -- a uniform 64-bit base in
        // scalar registers, one 32-bit lane offset per row fragment, the piece T in the instruction's scalar offset:
        //   * no 64-bit lane arithmetic (its zero high word is a register hipcc keeps across the whole kernel, spills, and reloads at
        //     the head of the epilogue -- a vmcnt(0) there);
        //   * a row past the tile's end gets an offset out of the resource's range: its loads return zeros without traffic and its
        //     stores are dropped -- no predication around 96 stores, no clamped re-reads;
        //   * everything lane-dependent comes from a FRESH lane id: loop-invariant addresses would be hoisted out of the tile loop,
        //     live through the chunk phases, and be spilled there.
        // Piece T of a lane: row-major forms: element (row) * D + 16 T + 4 g.  Images (kernels.h; whole fragments only -- the launcher
        // guarantees M % 16 == 0): fp32: fragment * 6144 + [(T >> 1) * 512 + (T & 1) * 128] + (g & 1) * 256 + (g >> 1) * 64 + 4 i;
        // bf16: fragment * 6144 + [(T >> 1) * 512 + (T & 1) * 256] + (g >> 1) * 128 + 8 i + 4 (g & 1)   ([..] = the scalar part).
        constexpr uint32_t OOB = 0x80000000u;
        // byte offsets of this lane's pieces: f32 selects the fp32 (x) or the bf16 (y1, xn) form, img the image or the row-major one
        auto lane_off = [&](int t_nrows, bool f32, bool img, uint32_t (&off)[2], int& g_) __attribute__((always_inline)) {
            int ln;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
            const int i_ = ln & 15;
            g_ = ln >> 4;
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int r = wave * 32 + mm * 16 + i_;
                const uint32_t e = img ? (uint32_t)((wave * 32 + mm * 16) * D) + (f32 ? (uint32_t)((g_ & 1) * 256 + (g_ >> 1) * 64 + 4 * i_)
                                                                                     : (uint32_t)((g_ >> 1) * 128 + 8 * i_ + 4 * (g_ & 1)))
                                       : (uint32_t)(r * D + 4 * g_);
                off[mm] = r < t_nrows ? e * (f32 ? 4u : 2u) : OOB;  // (scaled first: OOB times 4 would wrap back into range)
            }
        };
        // all four lanes of a row (l ^ 16, l ^ 32) get the same sum, added in the same order (no LDS crossbar, no lane-id register)
        auto row_sum = [&](float v) __attribute__((always_inline)) -> float {
#pragma clang fp contract(off)
            const uint32_t u = __builtin_bit_cast(uint32_t, v);
            const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            const float a = __builtin_bit_cast(float, (uint32_t)s16[0]) + __builtin_bit_cast(float, (uint32_t)s16[1]);
            const uint32_t ua = __builtin_bit_cast(uint32_t, a);
            const auto s32 = __builtin_amdgcn_permlane32_swap(ua, ua, false, false);
            return __builtin_bit_cast(float, (uint32_t)s32[0]) + __builtin_bit_cast(float, (uint32_t)s32[1]);
        };
        constexpr int XI_T1 = 512, XI_T0 = 128, YI_T1 = 512, YI_T0 = 256;  // image pieces: (T >> 1) * _T1 + (T & 1) * _T0 elements
#define PIECE_X(img, T) ((img) ? ((T) >> 1) * XI_T1 + ((T) & 1) * XI_T0 : 16 * (T))
#define PIECE_Y(img, T) ((img) ? ((T) >> 1) * YI_T1 + ((T) & 1) * YI_T0 : 16 * (T))

        // ---- row phase: v = x + y1 in the ACCUMULATOR layout; LayerNorm-2 there (a row's four g-lanes hold all of it: two cross-lane
        // steps); the normalised values packed to bf16 ARE the fc1 operand (the fc1 weight image lists k in the accumulator's column
        // order: mlp16_pack_kernel); and v seeds the fc2 accumulators, so the epilogue never re-reads x and y1.
        // All 96 loads of a lane (36 KiB per wave) are in flight together: one memory latency per tile.
        {
#pragma clang fp contract(off)
            uint32_t xo[2], yo[2];
            uint32_t g2base;  // LN-2 gamma in accumulator column order: gam[16 T + 4 g ..] at + 64 T bytes (beta: + D * 4)
            uint32_t bpbase = 0;  // (FOLD) proj bias, the same way
            {
                int g_;
                lane_off(nrows, true, XIN, xo, g_);
                lane_off(nrows, false, IMG, yo, g_);
                if constexpr (FOLD) {
                    // the attention rows as the proj product's B operand: lane (g, i) = row i, chunk 4 ks + g (8 columns = 16 bytes) of k-step ks.
                    // Image: fragment base + chunk * 256 + 16 i (1 KiB per wave instruction); row-major: row * 768 + 16 chunk.
                    int ln;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                    const int i_ = ln & 15;
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const int r = wave * 32 + mm * 16 + i_;
                        const uint32_t e = IMG ? (uint32_t)((wave * 32 + mm * 16) * D * 2 + g_ * 256 + i_ * 16) : (uint32_t)(r * D * 2 + g_ * 16);
                        yo[mm] = r < nrows ? e : OOB;
                    }
                    bpbase = (uint32_t)(uintptr_t)(LDS_AS char*)bps + 16 * g_;
                }
            }
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)row0 * D), 0, nrows * D * 4, 0x00020000);
            // (no y1: an empty range -- every piece reads as zero)
            const __amdgpu_buffer_rsrc_t ry =
                __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.y1 + (int64_t)row0 * D), 0, p.y1 ? nrows * D * 2 : 0, 0x00020000);
            f32x4 xv[2][NT16];
            u32x2 yv[2][NT16];
#pragma unroll
            for (int T = 0; T < NT16; ++T)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) xv[mm][T] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo[mm], PIECE_X(XIN, T) * 4, XIN ? AUX_ROWS : 0));
            if constexpr (FOLD) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) X[mm][ks] = __builtin_amdgcn_raw_buffer_load_b128(ry, yo[mm], ks * (IMG ? 1024 : 64), IMG ? AUX_ROWS : 0);
            } else {
#pragma unroll
                for (int T = 0; T < NT16; ++T)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) yv[mm][T] = __builtin_amdgcn_raw_buffer_load_b64(ry, yo[mm], PIECE_Y(IMG, T) * 2, IMG ? AUX_ROWS : 0);
            }
            if (tid == 0) {
                asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt_req) : "memory");
                if (nt_req == last_fetch) *p.counter = 0;
            }
            if constexpr (FOLD) {
                // the fc2 accumulators start from x + b_proj, take att Wproj^T through six proj phases, and come out as v = x + proj(att) + b
                sfor<0, NT16 / 4>([&](auto Q_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int T0 = 4 * decltype(Q_)::value;
                    f32x4 bb[4];
                    const uint32_t ba = bpbase;
                    f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                    DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, T0 * 64, T0 * 64 + 64, T0 * 64 + 128, T0 * 64 + 192);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm) {
                            const f32x4 c = xv[mm][T0 + t];
                            const f32x2 a = f32x2{c[0], c[1]} + f32x2{bb[t][0], bb[t][1]};
                            const f32x2 b = f32x2{c[2], c[3]} + f32x2{bb[t][2], bb[t][3]};
                            f32x4 sd = {a[0], a[1], b[0], b[1]};
                            asm volatile("" : "+a"(sd));
                            acc2[T0 + t][mm] = sd;
                        }
                    __builtin_amdgcn_sched_barrier(0);
                });
                rd_frag(I0{}, I0{}, fbase + (cons % 3) * UNIT);
                phase(TP{}, std::integral_constant<int, 0>{}, IM1{}, I0{}, I0{}, 0);
                phase(TP{}, std::integral_constant<int, 1>{}, IM1{}, I0{}, I0{}, 0);
                phase(TP{}, std::integral_constant<int, 2>{}, IM1{}, I0{}, I0{}, 0);
                phase(TP{}, std::integral_constant<int, 3>{}, IM1{}, I0{}, I0{}, 0);
                phase(TP{}, std::integral_constant<int, 4>{}, IM1{}, I0{}, I0{}, 0);
                phase(TP{}, std::integral_constant<int, 5>{}, IM1{}, I0{}, IM1{}, 0);
            }
            // (packed fp32 arithmetic: no MFMA runs beside the row phases, and it halves their vector instructions)
            float mean[2], rstd[2];
            sfor<0, 2>([&](auto M_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int mm = decltype(M_)::value;
                f32x2 rs2 = {0.f, 0.f};
#pragma unroll
                for (int T = 0; T < NT16; ++T) {
                    f32x2 a, b;
                    if constexpr (FOLD) {
                        const f32x4 c = acc2[T][mm];  // v = x + proj(att) + b_proj, read back from the accumulator file
                        a = f32x2{c[0], c[1]};
                        b = f32x2{c[2], c[3]};
                    } else {
                        const bf16x4 y = __builtin_bit_cast(bf16x4, yv[mm][T]);
                        a = f32x2{xv[mm][T][0], xv[mm][T][1]};
                        b = f32x2{xv[mm][T][2], xv[mm][T][3]};
                        a = a + f32x2{(float)y[0], (float)y[1]};
                        b = b + f32x2{(float)y[2], (float)y[3]};
                    }
                    rs2 = rs2 + a;
                    rs2 = rs2 + b;
                    xv[mm][T] = f32x4{a[0], a[1], b[0], b[1]};
                }
                mean[mm] = row_sum(rs2[0] + rs2[1]) * (1.0f / D);
                const f32x2 mean2 = {mean[mm], mean[mm]};
                f32x2 qs2 = {0.f, 0.f};
#pragma unroll
                for (int T = 0; T < NT16; ++T) {
                    const f32x2 a = f32x2{xv[mm][T][0], xv[mm][T][1]} - mean2, b = f32x2{xv[mm][T][2], xv[mm][T][3]} - mean2;
                    qs2 = __builtin_elementwise_fma(a, a, qs2);
                    qs2 = __builtin_elementwise_fma(b, b, qs2);
                }
                rstd[mm] = 1.0f / sqrtf(row_sum(qs2[0] + qs2[1]) * (1.0f / D) + p.ln_eps);
            });
            {   // (re-derived here from a fresh lane id: computed with the other offsets at the head of the row phase it lived across the six proj phases,
                //  and hipcc spilled it there and reloaded it here -- a scratch load, whose vmcnt(0) also waits for the weight stream)
                int ln;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
                g2base = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 16 * (ln >> 4);
            }
            sfor<0, NKS>([&](auto S_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int ks = decltype(S_)::value;
                f32x4 g0, g1, b0, b1_;  // gamma / beta of tiles 2 s, 2 s + 1 at this lane's columns
                const uint32_t ga = g2base;
                DSR128X4_WAIT(g0, g1, b0, b1_, ga, ks * 128, ks * 128 + 64, D * 4 + ks * 128, D * 4 + ks * 128 + 64);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const f32x2 mean2 = {mean[mm], mean[mm]}, rstd2 = {rstd[mm], rstd[mm]};
                    uint32_t w[4];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const f32x4 v = xv[mm][2 * ks + t], gq = t ? g1 : g0, bq_ = t ? b1_ : b0;
                        const f32x2 ya = __builtin_elementwise_fma((f32x2{v[0], v[1]} - mean2) * rstd2, f32x2{gq[0], gq[1]}, f32x2{bq_[0], bq_[1]});
                        const f32x2 yb = __builtin_elementwise_fma((f32x2{v[2], v[3]} - mean2) * rstd2, f32x2{gq[2], gq[3]}, f32x2{bq_[2], bq_[3]});
                        w[2 * t] = pack_bf16x2(ya[0], ya[1]);
                        w[2 * t + 1] = pack_bf16x2(yb[0], yb[1]);
                        if constexpr (!FOLD) {  // (FOLD: the accumulators already hold v)
                            f32x4 sd = v;
                            asm volatile("" : "+a"(sd));  // the seed goes to the accumulator file at once
                            acc2[2 * ks + t][mm] = sd;
                        }
                    }
                    X[mm][ks] = u32x4{w[0], w[1], w[2], w[3]};  // k-slots j < 4: tile 2 s, j >= 4: tile 2 s + 1
                }
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
            f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
            asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(q3));
        }
        // tail: second eight of the last half 1, then B1(last); its prefetch is the next tile's A0(0)
        sfor<8, 16>([&](auto U_) __attribute__((always_inline)) { gelu_unit(I1{}, U_); });
        __builtin_amdgcn_sched_barrier(0);
        phase(TB{}, I1{}, IM1{}, I0{}, IM1{}, 0);
        PSTAMP(3);
        PSTAMP_CLK(1);

        // ---- epilogue: x <- acc2 + b2 (acc2 started from v = x + y1: nothing is re-read).  This workgroup owns its rows: in place.
        //      (row-major in, image out -- the first block of a forward -- converts in place: a wave's 32 rows are the bytes of its two
        //       fragments, and every old value was loaded in the row phase)
        {
#pragma clang fp contract(off)
            uint32_t xso[2], nso[2];
            uint32_t b2base, g1base;
            {
                int g_;
                lane_off(nrows, true, IMG, xso, g_);
                lane_off(nrows, false, IMG, nso, g_);
                b2base = (uint32_t)(uintptr_t)(LDS_AS char*)b2s + 16 * g_;   // b2[16 T + 4 g ..]: + 64 T bytes
                g1base = (uint32_t)(uintptr_t)(LDS_AS char*)gam1 + 16 * g_;  // next LN-1 gamma (beta: + D * 4)
            }
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)row0 * D), 0, nrows * D * 4, 0x00020000);
            f32x2 rs2[2] = {{0.f, 0.f}, {0.f, 0.f}};
            sfor<0, NT16 / 4>([&](auto Q_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int T0 = 4 * decltype(Q_)::value;
                f32x4 bb[4];
                const uint32_t ba = b2base;
                f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, T0 * 64, T0 * 64 + 64, T0 * 64 + 128, T0 * 64 + 192);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const f32x4 c = acc2[T0 + t][mm];
                        const f32x2 a = f32x2{c[0], c[1]} + f32x2{bb[t][0], bb[t][1]};
                        const f32x2 b = f32x2{c[2], c[3]} + f32x2{bb[t][2], bb[t][3]};
                        rs2[mm] = rs2[mm] + a;
                        rs2[mm] = rs2[mm] + b;
                        const f32x4 v = {a[0], a[1], b[0], b[1]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rx, xso[mm], PIECE_X(IMG, T0 + t) * 4, IMG ? AUX_ROWS : 0);
                        acc2[T0 + t][mm] = v;
                    }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (p.xn_out) {
                // LayerNorm-1 of the next block on the finished rows (the four g-lanes of a row hold all of it), as bf16
                float mean[2], rstd[2];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    mean[mm] = row_sum(rs2[mm][0] + rs2[mm][1]) * (1.0f / D);
                    const f32x2 mean2 = {mean[mm], mean[mm]};
                    f32x2 qs2 = {0.f, 0.f};
#pragma unroll
                    for (int T = 0; T < NT16; ++T) {
                        const f32x2 a = f32x2{acc2[T][mm][0], acc2[T][mm][1]} - mean2, b = f32x2{acc2[T][mm][2], acc2[T][mm][3]} - mean2;
                        qs2 = __builtin_elementwise_fma(a, a, qs2);
                        qs2 = __builtin_elementwise_fma(b, b, qs2);
                    }
                    rstd[mm] = 1.0f / sqrtf(row_sum(qs2[0] + qs2[1]) * (1.0f / D) + p.ln_eps);
                }
                const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)p.xn_out + (int64_t)row0 * D), 0, nrows * D * 2, 0x00020000);
                sfor<0, NT16 / 2>([&](auto S_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int T0 = 2 * decltype(S_)::value;
                    f32x4 g0, g1, b0, b1_;
                    const uint32_t ga = g1base;
                    DSR128X4_WAIT(g0, g1, b0, b1_, ga, T0 * 64, T0 * 64 + 64, D * 4 + T0 * 64, D * 4 + T0 * 64 + 64);
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm) {
                            const f32x2 mean2 = {mean[mm], mean[mm]}, rstd2 = {rstd[mm], rstd[mm]};
                            const f32x4 v = acc2[T0 + t][mm], gq = t ? g1 : g0, bq_ = t ? b1_ : b0;
                            const f32x2 ya = __builtin_elementwise_fma((f32x2{v[0], v[1]} - mean2) * rstd2, f32x2{gq[0], gq[1]}, f32x2{bq_[0], bq_[1]});
                            const f32x2 yb = __builtin_elementwise_fma((f32x2{v[2], v[3]} - mean2) * rstd2, f32x2{gq[2], gq[3]}, f32x2{bq_[2], bq_[3]});
                            u32x2 o2;
                            o2[0] = pack_bf16x2(ya[0], ya[1]);
                            o2[1] = pack_bf16x2(yb[0], yb[1]);
                            __builtin_amdgcn_raw_buffer_store_b64(o2, rn, nso[mm], PIECE_Y(IMG, T0 + t) * 2, IMG ? AUX_ROWS : 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
#undef PIECE_X
#undef PIECE_Y
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

bool hipt_mlp16_supported(int dtype, int D_, int hidden) {
    return dtype == HIPT_BF16 && D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536;
}

size_t hipt_mlp16_packed_bytes(int D_, int hidden, bool with_proj) { return (size_t)2 * D_ * hidden * 2 + (with_proj ? (size_t)D_ * D_ * 2 : 0); }

// wproj != null: format 3 (six proj units in front of the pass), else format 2
int hipt_mlp16_pack_launch(const void* w1, const void* w2, int D_, int hidden, void* packed, hipStream_t st, const void* wproj) {
    if (!(D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536)) {
        hipt_set_error("mlp16 pack: unsupported D=%d hidden=%d", D_, hidden);
        return HIPT_E_UNSUPPORTED;
    }
    const int64_t chunks = (int64_t)((hidden / 128) * 4 + (wproj ? 6 : 0)) * (UNIT / 16);
    hipLaunchKernelGGL(mlp16_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w1, (const bf16_t*)w2, (const bf16_t*)wproj, hidden,
                       (u32x4*)packed);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

constexpr int MLP16_FMT = 2, MLP16_FMT_FOLD = 3;  // (include/hipt_abmil.h: hipt_block_weights.mlp_pk_fmt; 3 = with the proj units in front)
template <int DBG>
int hipt_mlp16_launch_dbg(const MlpParams& p_in, hipStream_t st) {
    MlpParams p = p_in;
    const bool fold = p.fold != 0;
    const int lds = 3 * UNIT + (3 * D + p.hidden) * 4 + 16 + 2 * D * 4 + 256 + (fold ? D * 4 : 0);
    if (!p.wpk || !(p.wpk_fmt == MLP16_FMT || p.wpk_fmt == MLP16_FMT_FOLD) || (p.img & 2 && !(p.img & 1)) || (p.img && p.M % 16 != 0) ||
        (fold && (p.wpk_fmt != MLP16_FMT_FOLD || !p.y1 || !p.bproj))) {
        hipt_set_error("mlp16: needs its packed weights; activation images need M %% 16 == 0 and img in {0, 1, 3}; proj folding needs image format 3, the attention "
                       "output in y1 and the proj bias (img=%d, M=%d, fold=%d, fmt=%d)", p.img, p.M, p.fold, p.wpk_fmt);
        return HIPT_E_BADARG;
    }
    if (!fold && p.wpk_fmt == MLP16_FMT_FOLD) p.wpk = (const char*)p.wpk + 6 * UNIT;  // (the MLP's own units lie behind the six proj units)
    auto k = fold ? (p.img == 3 ? mlp16_kernel<true, true, DBG, true> : p.img == 1 ? mlp16_kernel<true, false, DBG, true> : mlp16_kernel<false, false, DBG, true>)
                  : (p.img == 3 ? mlp16_kernel<true, true, DBG> : p.img == 1 ? mlp16_kernel<true, false, DBG> : mlp16_kernel<false, false, DBG>);
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)mlp16_kernel<true, true, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp16_kernel<true, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp16_kernel<false, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp16_kernel<true, true, DBG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp16_kernel<true, false, DBG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp16_kernel<false, false, DBG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(mlp16) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("mlp16: cannot query the device");
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
    // (round 6, one box, tools/streams_by_regions_bench.py: cutting the leftover tiles of LONG launches too is 1.0-1.9 % slower at every call size from 4 to
    //  24 regions, on one stream and on several: 128 workgroups each pay a pass over the weights for 16 rows)
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
        hipt_set_error("mlp16: hipMemsetAsync(counter) failed");
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
        fprintf(stderr, "[mlp16 dbg=%d hidden=%d grid=%d tiles=%d(+%d)] total %.1f us | tile %d of each workgroup: rows+LN %.1f, chunks %.1f (%.2f GHz), epilogue %.1f\n",
                DBG, p.hidden, grid, p.full_tiles, p.ntiles - p.full_tiles, (double)(t4 - t0) * 0.01, PSTAMP_SEQ, pro, chunks, ghz, epi);
    }
#endif
    return HIPT_OK;
}

int hipt_mlp16_launch(const MlpParams& p, hipStream_t st) { return hipt_mlp16_launch_dbg<0>(p, st); }

