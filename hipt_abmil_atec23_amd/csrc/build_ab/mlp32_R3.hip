// ROUND 3 FORM, kept for A/B runs (tools/mlp_probe.hip -DPROBE_32R3); the library builds csrc/mlp32.hip.
// FUSED MLP SUB-BLOCK for D = 384 (ViT-256) on 32x32x16 MFMAs:   x <- x + y1 + fc2( GELU( fc1( LN2(x + y1) ) ) )
//   (Block.forward second half, HIPT_4K/vision_transformer.py:151 with Mlp.forward :98-104.)
//
// Same data flow, ring protocol and phase order as mlp_pipe.hip (one 4-wave workgroup owns 128 rows; LN2(x+y1) as MFMA operand
// fragments, fc1 half-chunk accumulators -> GELU -> re-packed in registers as the fc2 operand, the [128, 384] fc2 accumulator:
// all in registers at one wave per SIMD; only weights stream through a 3 x 48 KiB LDS-DMA ring; phases A0(c) B1(c-1) A1(c)
// B0(c) so that a half's GELU hides under the two phases that follow its fc1).  What changes is the MFMA shape:
//   * v_mfma_f32_32x32x16_bf16 instead of 16x16x32.  The kernel is ISSUE-bound, not MFMA-bound: per 128 MFMA cycles a wave
//     also has to issue ~70 cycles of GELU arithmetic, ~70 of LDS-DMA pieces and its fragment reads, and a 16x16x32 MFMA
//     holds the SIMD's vector issue for 8 of its 16 cycles (64 of 128 left), a 32x32x16 for 8 of its 32 (96 of 128 left).
//   * a wave's 32 rows are ONE B operand (column = row): lane l = 32 h + 16 m + li holds row (fragment m, li) and, per 16-deep
//     k-step, 8 k values of half h.  The row phases still load / normalise in the 16-row fragment layout of the activation
//     images (lane (li, g) owns chunks g + 4c of BOTH fragments); twelve v_permlane16_swap per chunk pair turn that into the
//     32-row operand: k-step 2c + p of lane half h carries k = 32 c + 16 h + 8 p + (0..7) -- the weight image is built for
//     exactly that order, so no data is moved for it.
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
constexpr int D = 384, NCH = 12, NKS = 24, NOT = 12, TMR = 128;   // NKS: 16-deep k-steps of fc1; NOT: 32-wide output tiles
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
#define PSTAMP(k)                                                                                                    \
    do {                                                                                                             \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

enum { KA = 0, KB = 1, KP = 2 };  // phase kind: fc1 half / fc2 half / (FOLD) proj: output tiles 2 H, 2 H + 1 of the attention branch

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
// The six proj units (FOLD), stored behind the fc1 / fc2 units: unit u = output tiles 2 u, 2 u + 1 of the attention branch, fragments as in
// an fc1 unit: fragment 4 gg + 2 p + U: element j = Wp[32 (2 u + U) + r][32 gg + 16 p + 8 (j >> 2) + 4 h + (j & 3)].
__global__ void mlp32_pack_proj_kernel(const bf16_t* __restrict__ wp, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte lane chunk
    if (i >= 6 * (UNIT / 16)) return;
    const int u = i / (UNIT / 16), o = i % (UNIT / 16), frag = o >> 6, lane = o & 63, r = lane & 31, h = lane >> 5;
    const int gg = frag >> 2, pp = (frag >> 1) & 1, U = frag & 1;
    const bf16_t* row = wp + (int64_t)(32 * (2 * u + U) + r) * D + 32 * gg + 16 * pp + 4 * h;
    const u32x2 lo = *(const u32x2*)row, hi = *(const u32x2*)(row + 8);
    out[i] = u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// DBG (tools/mlp_probe.hip only): 1 = no weight DMA / ring syncs, 2 = GELU replaced by a plain pack, 4 = no MFMAs, 8 = no LDS
// fragment reads.
// FOLD: the attention branch's proj Linear runs here too (p.y1 = the attention output image [M, 384] bf16, six more weight units, p.bproj):
// see the row phase below.
template <bool IMG = false, bool XIN = false, int DBG = 0, bool FOLD = false>
__global__ __launch_bounds__(256, 1) void mlp32_kernel(const MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gam = (float*)(smem + 3 * UNIT);
    float* bet = gam + D;
    float* b2s = bet + D;
    float* b1s = b2s + D;                  // [hidden]
    int* tile_s = (int*)(b1s + p.hidden);  // [2] tile handed to this workgroup, double-buffered by parity
    float* gam1 = (float*)(tile_s + 4);    // next block's LayerNorm-1 (gamma | beta), if p.xn_out
    float* bps = gam1 + 2 * D;             // proj bias (FOLD)
    float* pfj = bps + D;                  // [64] where the L2-prefetch loads below drop their dwords (never read)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;        // the row phases' 16-row fragment view: lane (li, g) owns chunks g + 4c
    const int h = lane >> 5, m = (lane >> 4) & 1;   // the MFMA view: lane = 32 h + 16 m + li holds row (fragment m, li), k half h
    const int nchunk = p.hidden / 128;
    const int upt_mlp = 4 * nchunk;               // fc1 / fc2 units of a tile pass
    const int upt = upt_mlp + (FOLD ? 6 : 0);     // ring units (phases) per tile pass: FOLD: six proj units first

    // ---- weight DMA: unit pos of the image = 48 pieces of 1 KiB, byte for byte what its ring slot holds; wave w issues pieces
    // 12 w .. 12 w + 11.  An LDS-DMA instruction takes its LDS base from M0, and it is WRITING M0 that makes a piece expensive
    // (tools/issue_mix_probe.hip: +36 cycles per piece with a new M0, +2 with the same M0 and the piece selected by the instruction's
    // immediate offset, which is added to the LDS and to the global address alike): four consecutive pieces share one M0.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (2 * p.hidden * D + (FOLD ? D * D : 0)) * 2, 0x00020000);
    const uint32_t ilane = (uint32_t)(12 * wave * 1024 + lane * 16);
    int ioff = 0, islot = 0, ipos = 0;
    auto set_issue = [&](int pos, int slot) {
        // (the image: fc1 / fc2 units in pass order, then the six proj units -- kernels without FOLD never see those)
        ioff = (FOLD ? (pos < 6 ? upt_mlp + pos : pos - 6) : pos) * UNIT;
        islot = slot;
    };
    auto dma_piece = [&](auto T_) __attribute__((always_inline)) {
        constexpr int t = decltype(T_)::value;
        if constexpr ((DBG & 1) == 0 && (DBG & 32) == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + islot * UNIT + (12 * wave + (t & ~3)) * 1024), 16, ilane, ioff + (t & ~3) * 1024, (t & 3) * 1024, 0);
    };

    // ---- L2 prefetch of a tile's row-phase inputs.  The row phases are latency: one wave per SIMD waits 4-5 us for 36 KiB from HBM,
    // twice before the first phase and again in the epilogue (13 + 13 us of a 97 us tile with the matrix pipes idle).  Touching one
    // dword of every 128-byte line from inside a chunk phase a few microseconds earlier turns those waits into L2 hits.  The loads are
    // LDS-DMA (no destination register to keep alive), all into one 256-byte scratch line; rows past the tile's end are out of the
    // resource's range and dropped.  Nine instructions per wave: line (4 k + wave) * 64 + lane of x (k < 6) and of y1 (k < 3).
    auto prefetch_rows = [&](int t_row0, int t_nrows) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)t_row0 * D), 0, t_nrows * D * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry =
            __builtin_amdgcn_make_buffer_rsrc((void*)((const bf16_t*)p.y1 + (int64_t)t_row0 * D), 0, p.y1 ? t_nrows * D * 2 : 0, 0x00020000);
        const uint32_t vo = (uint32_t)(wave * 8192 + lane * 128);
#pragma unroll
        for (int k = 0; k < 6; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (LDS_AS void*)pfj, 4, vo, k * 32768, 0, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (LDS_AS void*)pfj, 4, vo, k * 32768, 0, 0);
    };
    constexpr int NRD = (DBG & 16) ? 2 : 4;  // fragment reads per group
    constexpr int NPF = 9;  // (the counted wait of a phase that prefetches)

    for (int i = tid; i < D; i += 256) {
        gam[i] = p.ln_w[i];
        bet[i] = p.ln_b[i];
        b2s[i] = p.b2[i];
        if (p.xn_out) {
            gam1[i] = p.ln_next_w[i];
            gam1[D + i] = p.ln_next_b[i];
        }
        if constexpr (FOLD) bps[i] = p.bproj[i];
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
    const uint32_t b1base = (uint32_t)(uintptr_t)(LDS_AS char*)b1s + 16 * h;          // b1[Hb + 32 U + 8 q + 4 h ..]: + (Hb + 32 U + 8 q) * 4
    const uint32_t tsbase = (uint32_t)(uintptr_t)(LDS_AS char*)tile_s;
    const uint32_t gbase = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 32 * g;           // (row phases: 16-row fragment view)
    const uint32_t b2base = (uint32_t)(uintptr_t)(LDS_AS char*)b2s + 16 * h;          // b2[32 O + 8 q + 4 h ..]: + (32 O + 8 q) * 4
    const uint32_t g1base = (uint32_t)(uintptr_t)(LDS_AS char*)gam1 + 16 * h;         // next LN-1 gamma (beta: + D * 4)
    const uint32_t g2base = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 16 * h;          // (FOLD) LN-2 gamma in accumulator column order (beta: + D * 4)
    const uint32_t bpbase = (uint32_t)(uintptr_t)(LDS_AS char*)bps + 16 * h;          // (FOLD) proj bias

    // ---- prime the ring: units 0 and 1 of the pass ----
    int cons = 0;  // units consumed since kernel start (slot = cons % 3)
    if (tile < p.ntiles) {
        set_issue(0, 0);
        sfor<0, 12>(dma_piece);
        set_issue(1, 1);  // its other ten pieces go out in groups 0..4 of the first phase, as in steady state
        sfor<0, 2>(dma_piece);
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
        PSTAMP(0);
        // next tile: requested now, handed to LDS behind the first row loads (the atomic's round trip is theirs too), read by every
        // wave after the first ring barrier
        int nt_req = 0;
        if (tid == 0) nt_req = atomicAdd(p.counter, 1);
        int tile_next = 0, row0_next = 0, nrows_next = 0;

        u32x4 X[NKS];  // the fc1 B operand: k-step s = 2 c + p, lane half h: columns 32 c + 16 p + 8 (j >> 2) + 4 h + (j & 3)

        f32x16 acc2[NOT];
#pragma unroll
        for (int o = 0; o < NOT; ++o)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[o][e] = 0.f;
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
        // NB/nb: the NEXT phase is an fc1 phase and starts from the bias at b1s offset nb (read with the cross-phase prefetch).
        //     NB = -1: last phase of the tile, nothing is prefetched (the row phases in between need the registers)
        //     PF: 1 = this phase also prefetches this tile's rows for the epilogue, 2 = the next tile's rows (groups 5 / 6, behind
        //     the phase's DMA pieces: the wait at group 11 leaves exactly the NPF prefetch loads in flight)
        auto phase = [&](auto KIND_, auto H_, auto GH_, auto GSEC_, auto NB_, int nb, auto PF_) __attribute__((always_inline)) {
            constexpr int kind = decltype(KIND_)::value, hh = decltype(H_)::value, gh = decltype(GH_)::value;
            constexpr int gsec = decltype(GSEC_)::value, needb = decltype(NB_)::value, pf = decltype(PF_)::value;
            const uint32_t sa = fbase + (cons % 3) * UNIT;
            const uint32_t sn = fbase + ((cons + 1) % 3) * UNIT;
            sfor<0, 12>([&](auto G_) __attribute__((always_inline)) {
                constexpr int gg = decltype(G_)::value, set = gg & 1;
                typedef std::integral_constant<int, set ^ 1> NS;
                // (1) fragment reads one group ahead
                if constexpr (gg == 11) {
                    if constexpr ((DBG & 1) == 0) {
                        if constexpr (pf != 0)
                            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPF) : "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of the next unit have landed
                        __builtin_amdgcn_s_barrier();                     // ... everyone's; unit cons-1 is no longer read
                    }
                    set_issue(ipos, (cons + 2) % 3);
                    ipos = ipos + 1 == upt ? 0 : ipos + 1;
                    if constexpr (needb < 0) {
                        LGKM(0);
                    } else {
                        rd_frag(NS{}, I0{}, sn);
                        if constexpr (needb > 0) {
                            bias_rd(nb);
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
                } else if constexpr (kind == KB) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) mma32<DBG>(acc2[gg], wA[set][t], hf[hh][t >> 1][t & 1]);
                } else {  // KP: as an fc1 group, into the (still idle) fc2 accumulators of output tiles 2 hh, 2 hh + 1
                    mma32<DBG>(acc2[2 * hh], wA[set][0], X[2 * gg]);
                    mma32<DBG>(acc2[2 * hh + 1], wA[set][1], X[2 * gg]);
                    mma32<DBG>(acc2[2 * hh], wA[set][2], X[2 * gg + 1]);
                    mma32<DBG>(acc2[2 * hh + 1], wA[set][3], X[2 * gg + 1]);
                }
                if constexpr (gg == 11) {
                    dma_piece(std::integral_constant<int, 0>{});
                    dma_piece(std::integral_constant<int, 1>{});
                } else if constexpr (gg <= 4) {
                    dma_piece(std::integral_constant<int, 2 + 2 * gg>{});
                    dma_piece(std::integral_constant<int, 3 + 2 * gg>{});
                }
                if constexpr (pf == 1 && gg == 5) prefetch_rows(row0, nrows);
                if constexpr (pf == 2 && gg == 5) prefetch_rows(row0_next, nrows_next);
                if constexpr (gh >= 0 && gg >= 4) {
                    gelu_unit(std::integral_constant<int, (gh >= 0 ? gh : 0)>{}, std::integral_constant<int, 8 * gsec + gg - 4>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            cons += 1;
        };
        typedef std::integral_constant<int, -1> IM1;
        typedef std::integral_constant<int, KA> TA;
        typedef std::integral_constant<int, KB> TB;
        typedef std::integral_constant<int, KP> TP;
#ifdef MLP32_PREFETCH  // (experiment, off: see prefetch_rows)
        typedef std::integral_constant<int, 1> PFA;
        typedef std::integral_constant<int, 2> PFB;
#else
        typedef I0 PFA;
        typedef I0 PFB;
#endif

        // 16-row fragments (chunk g + 4 c of fragments 0 / 1 per lane) -> the 32-row B operand X
        auto to_operand = [&](u32x4 (&af)[2][NCH]) __attribute__((always_inline)) {
            // 16-row fragments -> the 32-row B operand.  Lane (li, g = 2 h + m) holds chunks 2 h + m + 4 c of BOTH fragments; it needs
        // fragment m only, chunks 2 h + 4 c (E) and 2 h + 1 + 4 c (O).  Lanes l and l ^ 16 (m = 0 / 1, same h) hold each other's
        // missing chunks: one v_permlane16_swap per dword (odd 16-lane rows of the first operand <-> even rows of the second)
        // leaves E in the first and O in the second for every lane.  k-step 2 c + p then carries k = 32 c + 16 h + 8 p + (0..7).
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            u32x4 e4, o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane16_swap(af[0][c][e], af[1][c][e], false, false);
                e4[e] = sw[0];
                o4[e] = sw[1];
            }
            // ... and lanes l and l ^ 32 trade 4-column groups, so that the operand's k order is the column order of an accumulator
            // tile (lane half h: columns 8 q + 4 h + (0..3) of every 32): after the 16-lane swap half h holds columns 16 h + (0..15)
            // of the 32 as e4 = [G, G + 1], o4 = [G + 2, G + 3] (G = 4 h, groups of 4 columns); it keeps its even groups and
            // takes the other half's: k-step 2 c = [G0 | G2] (h = 0) / [G1 | G3] (h = 1), k-step 2 c + 1 = [G4 | G6] / [G5 | G7].
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
        };
        if constexpr (!FOLD) {
            // ---- activations: v = x + y1 -> LN2 -> operand fragments (16-row fragment view, as mlp_pipe.hip) ----
            u32x4 af[2][NCH];
#pragma unroll
            for (int mf = 0; mf < 2; ++mf) {
                int r = (wave * 2 + mf) * 16 + li;
                r = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
                // image forms: whole fragments only (the launcher guarantees M % 16 == 0); a fragment past the tile's end
                // re-reads fragment 0 of the tile (never stored)
                const int fr = (wave * 2 + mf) * 16 < nrows ? (wave * 2 + mf) * 16 : 0;
                // x: row-major: row r, floats (g + 4c) * 8 + 4hh;  image: fragment base + c * 512 + hh * 256 + lane * 4
                const float* xr = XIN ? p.x + (int64_t)(row0 + fr) * D + lane * 4 : p.x + (int64_t)(row0 + r) * D + g * 8;
                constexpr int xc_ = XIN ? 512 : 32, xh_ = XIN ? 256 : 4;
                f32x4 v[NCH][2];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    v[c][0] = *(const f32x4*)(xr + c * xc_);
                    v[c][1] = *(const f32x4*)(xr + c * xc_ + xh_);
                }
                if (p.y1) {
                    // y1 (bf16): row-major: row r, elements (g + 4c) * 8;  image: fragment base + c * 512 + lane * 8
                    const bf16_t* yr = IMG ? (const bf16_t*)p.y1 + (int64_t)(row0 + fr) * D + lane * 8 : (const bf16_t*)p.y1 + (int64_t)(row0 + r) * D + g * 8;
                    constexpr int yc_ = IMG ? 512 : 32;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const bf16x8 y = __builtin_bit_cast(bf16x8, *(const u32x4*)(yr + c * yc_));
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[c][0][e] += (float)y[e];
                            v[c][1][e] += (float)y[4 + e];
                        }
                    }
                }
                if (HIPT_STAMPS_ON(p.stamps)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (mf == 0) { PSTAMP(5); } else { PSTAMP(7); }
                }
                if (mf == 0 && tid == 0) {
                    asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt_req) : "memory");
                    if (nt_req == last_fetch) *p.counter = 0;
                }
                ln_rows_lds<D, NCH>(v, gbase, p.ln_eps, af[mf]);
                if (mf == 0) PSTAMP(6);
                if (mf == 0) {
                    // park the finished fragment in the accumulator file (idle during the row phase) while the other one
                    // is loaded and normalised: left alone, hipcc sends it to scratch and the reloads stall the first phase
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        u32x4& a2 = af[0][c];
                        asm volatile("" : "+a"(a2));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            PSTAMP(12);
            to_operand(af);
        } else {
            if (tid == 0) {
                asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt_req) : "memory");
                if (nt_req == last_fetch) *p.counter = 0;
            }
            // ---- FOLD: y1 = proj(att) is computed here instead of read.  (1) the attention output tile (bf16 image) is itself
            // an operand: its 16-byte chunks are the fragment layout; (2) six phases on the proj units into acc2 (idle until the
            // first fc2 phase): acc2[O][4 q + e] = column 32 O + 8 q + 4 h + e of this lane's row; (3) v = acc2 + b_proj + x becomes
            // the residual stream (stored as its image: the epilogue re-reads it instead of x and y1) and, LayerNorm-2'd, the fc1
            // operand -- whose k order IS the accumulator's column order.
            {
                u32x4 af[2][NCH];
#pragma unroll
                for (int mf = 0; mf < 2; ++mf) {
                    const int fr = (wave * 2 + mf) * 16 < nrows ? (wave * 2 + mf) * 16 : 0;
                    const bf16_t* yr = (const bf16_t*)p.y1 + (int64_t)(row0 + fr) * D + lane * 8;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) af[mf][c] = *(const u32x4*)(yr + c * 512);
                }
                to_operand(af);
            }
            // the residual rows of this tile, in the accumulator's layout: output tiles 0..5 are requested now and land under the proj
            // phases (their 96 registers are free until the first fc1 phase), tiles 6..11 when those are being added
            constexpr int xlo_ = XIN ? 512 : 32, xlq_ = XIN ? 64 : 8;
            // (pointers are formed where they are used, from values that are live anyway: kept across the phases they would be spilled,
            //  and a scratch reload inside a ring phase waits for the LDS-DMA in flight)
            auto x_ptrs = [&](const float*& xl, float*& xs, bool& live) __attribute__((always_inline)) {
                int li2 = li;
                asm volatile("" : "+v"(li2));
                const int r = wave * 32 + m * 16 + li2;
                live = r < nrows;
                const int frr = wave * 32 + m * 16;
                const int64_t rb = (int64_t)(row0 + (live ? r : 0)) * D + 4 * h, fb = (int64_t)(row0 + (live ? frr : 0)) * D;
                xl = XIN ? p.x + fb + 256 * h + 4 * li2 : p.x + rb;
                xs = p.x + fb + 256 * h + 4 * li2;
            };
            f32x4 xa[6][4], xb[6][4];
            {
                const float* xl;
                float* xs;
                bool live;
                x_ptrs(xl, xs, live);
#pragma unroll
                for (int O = 0; O < 6; ++O)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xa[O][q] = *(const f32x4*)(xl + xlo_ * O + xlq_ * q);
            }
            rd_frag(I0{}, I0{}, fbase + (cons % 3) * UNIT);
            phase(TP{}, I0{}, IM1{}, I0{}, I0{}, 0, I0{});
            phase(TP{}, I1{}, IM1{}, I0{}, I0{}, 0, I0{});
            phase(TP{}, std::integral_constant<int, 2>{}, IM1{}, I0{}, I0{}, 0, I0{});
            phase(TP{}, std::integral_constant<int, 3>{}, IM1{}, I0{}, I0{}, 0, I0{});
            phase(TP{}, std::integral_constant<int, 4>{}, IM1{}, I0{}, I0{}, 0, I0{});
            phase(TP{}, std::integral_constant<int, 5>{}, IM1{}, I0{}, IM1{}, 0, I0{});
            {
#pragma clang fp contract(off)
                const float* xl;
                float* xs;
                bool live;
                x_ptrs(xl, xs, live);
                // every old value of the tile is loaded (and waited for) before the first store: converting in place (row-major in,
                // image out: the first block of a forward) a lane's stores land where OTHER lanes' loads read
#pragma unroll
                for (int O = 0; O < 6; ++O)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xb[O][q] = *(const f32x4*)(xl + xlo_ * (6 + O) + xlq_ * q);
                sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int O = decltype(O_)::value;
                    f32x4 bb[4];
                    const uint32_t ba = bpbase;
                    f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                    DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
                    f32x16 t = acc2[O];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) t[4 * q + e] = (t[4 * q + e] + bb[q][e]) + (O < 6 ? xa[O < 6 ? O : 0][q][e] : xb[O < 6 ? 0 : O - 6][q][e]);
                    asm volatile("" : "+a"(t));  // back to the accumulator file at once (left to hipcc, the sums go to scratch)
                    acc2[O] = t;
                    __builtin_amdgcn_sched_barrier(0);
                });
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (one output tile at a time from here on: the accumulators live in the accumulator file, arithmetic needs them in arch
                //  VGPRs, and hipcc, left alone, fetches all 192 at once and spills)
                float rs = 0.f;
                sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int O = decltype(O_)::value;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc2[O][4 * q], acc2[O][4 * q + 1], acc2[O][4 * q + 2], acc2[O][4 * q + 3]};
                        if (live) *(f32x4*)(xs + 512 * O + 64 * q) = v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) rs += v[e];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                rs += __shfl_xor(rs, 32, 64);
                const float mean = rs * (1.0f / D);
                float qs = 0.f;
                sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
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
                sfor<0, NOT>([&](auto O_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int O = decltype(O_)::value;
                    f32x4 gq[4], bqv[4];
                    const uint32_t ga = g2base;
                    f32x4 &g0 = gq[0], &g1 = gq[1], &g2 = gq[2], &g3 = gq[3], &b0 = bqv[0], &b1_ = bqv[1], &b2_ = bqv[2], &b3 = bqv[3];
                    DSR128X4_WAIT(g0, g1, g2, g3, ga, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
                    DSR128X4_WAIT(b0, b1_, b2_, b3, ga, D * 4 + O * 128, D * 4 + O * 128 + 32, D * 4 + O * 128 + 64, D * 4 + O * 128 + 96);
                    uint32_t w[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc2[O][4 * q + e] - mean) * rstd, gq[q][e], bqv[q][e]);
                        w[2 * q] = pack_bf16x2(y[0], y[1]);
                        w[2 * q + 1] = pack_bf16x2(y[2], y[3]);
                    }
                    X[2 * O] = u32x4{w[0], w[1], w[2], w[3]};      // k-step 2 O: columns 8 q + 4 h + e of the 32, q = 0, 1
                    X[2 * O + 1] = u32x4{w[4], w[5], w[6], w[7]};  // k-step 2 O + 1: q = 2, 3
                    __builtin_amdgcn_sched_barrier(0);
                });
#pragma unroll
                for (int o = 0; o < NOT; ++o)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc2[o][e] = 0.f;
            }
        }
        PSTAMP(2);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memtime();

#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) hf[a][b][c] = u32x4{0u, 0u, 0u, 0u};
        // first fragments and bias of the pass (asm reads land asynchronously: nothing but the first phase may sit
        // between them and their counted wait -- in particular not the row phases, where the compiler moves registers)
        rd_frag(I0{}, I0{}, fbase + (cons % 3) * UNIT);
        bias_rd(0);
        // chunk 0 (peeled: no runtime branches around phases inside the steady-state loop).  Its half-0 GELUs have
        // only A1(0) to hide in: the second eight run bare.
        phase(TA{}, I0{}, IM1{}, I0{}, I1{}, 64, I0{});
        phase(TA{}, I1{}, I0{}, I0{}, I0{}, 0, I0{});
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
        phase(TB{}, I0{}, I1{}, I0{}, I1{}, 128, I0{});
        for (int c = 1; c < nchunk - 1; ++c) {
            phase(TA{}, I0{}, I1{}, I1{}, I0{}, 0, I0{});             // A0(c)   + second eight GELUs of half 1 of chunk c-1
            phase(TB{}, I1{}, I0{}, I0{}, I1{}, c * 128 + 64, I0{});  // B1(c-1) + first eight of half 0 of chunk c
            phase(TA{}, I1{}, I0{}, I1{}, I0{}, 0, I0{});             // A1(c)   + second eight of half 0
            phase(TB{}, I0{}, I1{}, I0{}, I1{}, (c + 1) * 128, I0{});  // B0(c) + first eight of half 1
        }
        {   // the last chunk (peeled): its first two phases also request the rows of the epilogue and of the next tile's row phase
            const int c = nchunk - 1;
            phase(TA{}, I0{}, I1{}, I1{}, I0{}, 0, PFA{});
            phase(TB{}, I1{}, I0{}, I0{}, I1{}, c * 128 + 64, PFB{});
            phase(TA{}, I1{}, I0{}, I1{}, I0{}, 0, I0{});
            phase(TB{}, I0{}, I1{}, I0{}, I1{}, 0, I0{});
        }
        LGKM(0);  // (the last B0 read a bias nobody uses: let it land before its registers are re-used ...
        {         //  ... and keep those registers allocated up to here: a fake use AFTER the wait)
            f32x4 &q0 = bq[0][0], &q1 = bq[0][1], &q2 = bq[0][2], &q3 = bq[0][3], &q4 = bq[1][0], &q5 = bq[1][1], &q6 = bq[1][2], &q7 = bq[1][3];
            asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7));
        }
        // tail: second eight of the last half 1, then B1(last); its prefetch is the next tile's A0(0)
        sfor<8, 16>([&](auto U_) __attribute__((always_inline)) { gelu_unit(I1{}, U_); });
        __builtin_amdgcn_sched_barrier(0);
        phase(TB{}, I1{}, IM1{}, I0{}, IM1{}, 0, I0{});
        PSTAMP(3);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime();

        // ---- epilogue: x <- x + y1 + acc2 + b2 (this workgroup owns its rows: in place, no other reader).
        //      Lane (h, m, li) holds row 32 w + 16 m + li; acc2[O][4 q + e] is output column 32 O + 8 q + 4 h + e.
        //      Images: chunk-of-8 index 4 O + q = g' + 4 c' with g' = q, c' = O, half h, image lane 16 q + li.
        {
#pragma clang fp contract(off)
            const int r = wave * 32 + m * 16 + li;
            const bool live = r < nrows;
            const int fr = wave * 32 + m * 16;  // (image forms: stored only when live, i.e. fr < nrows)
            const int64_t rb = (int64_t)(row0 + (live ? r : 0)) * D + 4 * h, fb = (int64_t)(row0 + (live ? fr : 0)) * D;
            // float / element offsets of piece (O, q): row-major rb + 32 O + 8 q; fp32 image fb + 512 O + 256 h + 64 q + 4 li;
            // bf16 image fb + 512 O + 128 q + 8 li + 4 h
            // (FOLD: the row phase left v = x + y1 where x was, as an image: that is what is re-read, and there is no y1)
            constexpr bool XI = XIN || FOLD;
            const float* xl = XI ? p.x + fb + 256 * h + 4 * li : p.x + rb;
            float* xs = IMG ? p.x + fb + 256 * h + 4 * li : p.x + rb;
            const bf16_t* yr = IMG ? (const bf16_t*)p.y1 + fb + 8 * li + 4 * h : (const bf16_t*)p.y1 + rb;
            constexpr int xlo_ = XI ? 512 : 32, xlq_ = XI ? 64 : 8, xso_ = IMG ? 512 : 32, xsq_ = IMG ? 64 : 8, yo_ = IMG ? 512 : 32, yq_ = IMG ? 128 : 8;
            float rs = 0.f;
            // three output tiles at a time: their old x (12 x 16 B) and y1 (12 x 8 B) pieces are requested one batch ahead.
            // Converting in place (row-major in, image out: the first block of a forward) a lane's stores land where OTHER lanes'
            // loads read: there every old value of the tile is loaded, and waited for, before the first store.
            f32x4 xv[XI != IMG ? NOT : 6][4];
            u32x2 yv[6][4];
            auto ld_batch = [&](auto B_) __attribute__((always_inline)) {
                constexpr int b = decltype(B_)::value, s0 = (b & 1) * 3;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int O = 3 * b + i;
                        if constexpr (XI == IMG) xv[s0 + i][q] = *(const f32x4*)(xl + xlo_ * O + xlq_ * q);
                        yv[s0 + i][q] = (!FOLD && p.y1) ? *(const u32x2*)(yr + yo_ * O + yq_ * q) : u32x2{0u, 0u};
                    }
            };
            if constexpr (XI != IMG) {
#pragma unroll
                for (int O = 0; O < NOT; ++O)
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[O][q] = *(const f32x4*)(xl + xlo_ * O + xlq_ * q);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            ld_batch(I0{});
            sfor<0, 4>([&](auto B_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                constexpr int b = decltype(B_)::value, s0 = (b & 1) * 3;
                if constexpr (b < 3) ld_batch(std::integral_constant<int, b + 1>{});
                sfor<0, 3>([&](auto I_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int i = decltype(I_)::value, O = 3 * b + i, xi = (XI != IMG) ? O : s0 + i;
                    f32x4 bb[4];
                    const uint32_t ba = b2base;
                    f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                    DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, O * 128, O * 128 + 32, O * 128 + 64, O * 128 + 96);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bf16x4 y = __builtin_bit_cast(bf16x4, yv[s0 + i][q]);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = ((acc2[O][4 * q + e] + bb[q][e]) + xv[xi][q][e]) + (float)y[e];
                        if (live) *(f32x4*)(xs + xso_ * O + xsq_ * q) = v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc2[O][4 * q + e] = v[e];
                            rs += v[e];
                        }
                    }
                });
            });
            if (p.xn_out) {
                // LayerNorm-1 of the next block on the finished row (the two h-lanes of a row hold all of it), as bf16
                rs += __shfl_xor(rs, 32, 64);
                const float mean = rs * (1.0f / D);
                float qs = 0.f;
#pragma unroll
                for (int O = 0; O < NOT; ++O)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float a = acc2[O][e] - mean;
                        qs = __builtin_fmaf(a, a, qs);
                    }
                qs += __shfl_xor(qs, 32, 64);
                const float rstd = 1.0f / sqrtf(qs * (1.0f / D) + p.ln_eps);
                bf16_t* nr = IMG ? (bf16_t*)p.xn_out + fb + 8 * li + 4 * h : (bf16_t*)p.xn_out + rb;
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
                        float y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = __builtin_fmaf((acc2[O][4 * q + e] - mean) * rstd, gq[q][e], bqv[q][e]);
                        u32x2 o2;
                        o2[0] = pack_bf16x2(y[0], y[1]);
                        o2[1] = pack_bf16x2(y[2], y[3]);
                        if (live) *(u32x2*)(nr + yo_ * O + yq_ * q) = o2;
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

bool hipt_mlp32_supported(int dtype, int D_, int hidden) {
    return dtype == HIPT_BF16 && D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536;
}

int hipt_mlp32_pack_launch(const void* w1, const void* w2, int D_, int hidden, void* packed, hipStream_t st, const void* wproj) {
    if (!(D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536)) {
        hipt_set_error("mlp32 pack: unsupported D=%d hidden=%d", D_, hidden);
        return HIPT_E_UNSUPPORTED;
    }
    const int64_t chunks = (int64_t)(hidden / 128) * 4 * (UNIT / 16);
    hipLaunchKernelGGL(mlp32_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const bf16_t*)w1, (const bf16_t*)w2, hidden, (u32x4*)packed);
    HIPT_CHECK_LAUNCH();
    if (wproj) {
        hipLaunchKernelGGL(mlp32_pack_proj_kernel, dim3((6 * (UNIT / 16) + 255) / 256), dim3(256), 0, st, (const bf16_t*)wproj, (u32x4*)packed + chunks);
        HIPT_CHECK_LAUNCH();
    }
    return HIPT_OK;
}

// (kernels.h declares the six-argument form since round 4)
int hipt_mlp32_pack_launch(const void* w1, const void* w2, int D_, int hidden, void* packed, hipStream_t st) {
    return hipt_mlp32_pack_launch(w1, w2, D_, hidden, packed, st, nullptr);
}

template <int DBG>
int hipt_mlp32_launch_dbg(const MlpParams& p_in, hipStream_t st) {
    MlpParams p = p_in;
    const int lds = 3 * UNIT + (3 * D + p.hidden) * 4 + 16 + 2 * D * 4 + D * 4 + 256;
    if (!p.wpk || p.wpk_fmt != 1 || (p.img & 2 && !(p.img & 1)) || (p.img && p.M % 16 != 0) || (p.fold && (!(p.img & 1) || !p.y1 || !p.bproj))) {
        hipt_set_error("mlp32: needs its packed weights; activation images need M %% 16 == 0 and img in {0, 1, 3}; fold needs images (img=%d, M=%d, fold=%d)", p.img,
                       p.M, p.fold);
        return HIPT_E_BADARG;
    }
#ifdef HIPT_EXPERIMENTS  // proj folded into the MLP (break-even): tools/mlp_probe.hip builds it, the library does not
    auto k = p.fold ? (p.img == 3 ? mlp32_kernel<true, true, DBG, true> : mlp32_kernel<true, false, DBG, true>)
             : p.img == 3 ? mlp32_kernel<true, true, DBG>
             : p.img == 1 ? mlp32_kernel<true, false, DBG>
                          : mlp32_kernel<false, false, DBG>;
#else
    if (p.fold) {
        hipt_set_error("mlp32: the proj-folding kernel exists only in experiment builds (HIPT_EXPERIMENTS)");
        return HIPT_E_UNSUPPORTED;
    }
    auto k = p.img == 3 ? mlp32_kernel<true, true, DBG> : p.img == 1 ? mlp32_kernel<true, false, DBG> : mlp32_kernel<false, false, DBG>;
#endif
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)mlp32_kernel<true, true, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp32_kernel<true, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp32_kernel<false, false, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess
#ifdef HIPT_EXPERIMENTS
            || hipFuncSetAttribute((const void*)mlp32_kernel<true, true, DBG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp32_kernel<true, false, DBG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess
#endif
        ) {
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
        double pro = 0, chunks = 0, epi = 0, ghz = 0, pp[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < grid; ++b) {
            pro += (double)(h[b * 16 + 2] - h[b * 16 + 0]) * 0.01 / grid;
            chunks += (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.01 / grid;
            epi += (double)(h[b * 16 + 4] - h[b * 16 + 3]) * 0.01 / grid;
            const int ix[6] = {0, 5, 6, 7, 12, 2};
            for (int i = 0; i < 5; ++i) pp[i] += (double)(h[b * 16 + ix[i + 1]] - h[b * 16 + ix[i]]) * 0.01 / grid;
            ghz += (double)(h[b * 16 + 9] - h[b * 16 + 8]) / (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.1 / grid;
        }
        fprintf(stderr, "[mlp32 dbg=%d hidden=%d grid=%d tiles=%d(+%d)] total %.1f us | tile %d of each workgroup: rows+LN %.1f, chunks %.1f (%.2f GHz), epilogue %.1f\n",
                DBG, p.hidden, grid, p.full_tiles, p.ntiles - p.full_tiles, (double)(t4 - t0) * 0.01, PSTAMP_SEQ, pro, chunks, ghz, epi);
        if (!p.fold) fprintf(stderr, "    rows+LN: loads 0 %.1f, LN 0 %.1f, loads 1 %.1f, LN 1 %.1f, to operand %.1f\n", pp[0], pp[1], pp[2], pp[3], pp[4]);
    }
#endif
    return HIPT_OK;
}

int hipt_mlp32_launch(const MlpParams& p, hipStream_t st) { return hipt_mlp32_launch_dbg<0>(p, st); }
