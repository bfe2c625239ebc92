// FUSED MLP SUB-BLOCK for D = 384 (ViT-256), software-pipelined:   x <- x + y1 + fc2( GELU( fc1( LN2(x + y1) ) ) )
//   (Block.forward second half, HIPT_4K/vision_transformer.py:151 with Mlp.forward :98-104.)
//
// Same data flow as mlp.hip (one 4-wave workgroup owns 128 rows; LN2(x+y1) as MFMA operand fragments, the fc1
// chunk accumulator, GELU re-packed in registers as the fc2 operand, the [128, 384] fc2 accumulator: all in
// registers at one wave per SIMD; only weights stream through LDS) but the per-chunk work is re-ordered so
// that the matrix pipe never waits for the vector pipe:
//   * a hidden chunk of 128 is cut in two halves.  Per chunk c the MFMA phases run in the order
//         A0(c)   fc1, hidden half 0          (96 MFMAs per wave, 3 weight slabs of 64 hidden x 128 k)
//         B1(c-1) fc2 of the PREVIOUS chunk's half 1
//         A1(c)   fc1, hidden half 1
//         B0(c)   fc2, half 0                 (3 slabs of 128 outputs x 64 hidden)
//     so GELU of half 0 (needed by B0(c)) has the two phases B1(c-1), A1(c) to hide in, and GELU of half 1
//     (needed by B1(c)) has B0(c), A0(c+1): one 2-element packed GELU (~70 VALU cycles) per 8-MFMA group,
//     inside the 8 of every 16 cycles in which an MFMA leaves the vector issue port free.
//   * one ring unit = one phase = 3 slabs (48 KiB); 3 units in LDS (being read / landed / landing).  One
//     barrier per phase; the LDS fragment reads run one 8-MFMA group ahead, across phase boundaries.
//   * the fc1 weight rows are laid into LDS in a permuted hidden order (the DMA picks the global row per LDS
//     row) chosen so that the 8 fc2 K slots a lane owns after GELU are 8 CONSECUTIVE hidden units: the fc2
//     weight fragment is then one plain 16-byte LDS read, exactly like the fc1 fragment.
//   * the fc1 bias enters as the C operand of a half's first MFMA (no add, no accumulator init).
//   * the fc2 weight rows (= output columns) are permuted the same way: a lane ends with 8 CONSECUTIVE output columns per
//     fragment pair -> 32-byte row pieces for the fp32 residual stream, 16-byte ones for bf16.
//   * optionally the epilogue also applies the NEXT block's LayerNorm-1 to the rows it has just finished and writes
//     them as bf16: the next QKV GEMM then needs neither the fp32 row load nor the LayerNorm (a third of its time).
//   * persistent workgroups pull tiles from an atomic counter; the weight stream is tile-independent and
//     runs continuously across tiles.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"
#include "mlp_common.h"
#include "pipe_common.h"

// activations are touched once per launch: loaded / stored with the non-temporal hint they do not push the weight image
// (re-read by every workgroup, every tile) out of the L2
#ifndef MLP_NT  // (measured: the hint makes the row phases 11 % SLOWER on MI355X -- kept as an experiment switch)
#define NT_LD(p) (*(p))
#define NT_ST(v, p) (*(p) = (v))
#else
#define NT_LD(p) __builtin_nontemporal_load(p)
#define NT_ST(v, p) __builtin_nontemporal_store(v, p)
#endif
#ifndef PSTAMP_SEQ
#define PSTAMP_SEQ 0  // which tile of a workgroup the debug stamps describe (0 = the first: every CU in step)
#endif
namespace {

constexpr int D = 384, NCH = 12, NF2 = 24, TMR = 128;
constexpr int SLAB = 16384, UNIT = 3 * SLAB;

#define PSTAMP(k)                                                                                                    \
    do {                                                                                                             \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

enum { KA = 0, KB = 1 };            // phase kind: fc1 half / fc2 half

// Where the weights of ring unit `pos` of a tile pass come from.  Positions: A0(0) A1(0) B0(0) | A0(c) B1(c-1) A1(c)
// B0(c) ... | B1(n-1); an A unit = 64 hidden rows x all k of W1 (slab j: k [128j, +128)), a B unit = all 384 output rows
// of W2 x 64 hidden (slab j: outputs [128j, +128)).  Byte offsets: piece (j, q) of the unit starts at
// base + j * joff + (q & 1) * q1 + (q >> 1) * q2 (+ the per-lane part).
struct UnitSrc {
    bool is_a;
    int base, joff, q1, q2;
};
__device__ __forceinline__ UnitSrc unit_src(int pos, int nchunk, int hidden) {
    const int upt = 4 * nchunk;
    bool is_a;
    int c, h;
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
    UnitSrc u;
    u.is_a = is_a;
    if (is_a) {  // hidden rows [128c + 64h, +64) x all k
        u.base = (c * 128 + 64 * h) * D * 2;
        u.joff = 128 * 2;
        u.q1 = 32 * D * 2;
        u.q2 = 64 * 2;
    } else {  // output rows [128j, +128) x hidden [128c + 64h, +64)
        u.base = (c * 128 + 64 * h) * 2;
        u.joff = 128 * hidden * 2;
        u.q1 = 32 * hidden * 2;
        u.q2 = 64 * hidden * 2;
    }
    return u;
}
// per-lane part: LDS row R = wave*8 + (lane>>3) + 32q of a slab holds matrix row U(R) (the fragment-pair permutation,
// see the kernel), and the lane's 16-byte chunk is its LDS chunk position XOR ((R>>1)&7)
__device__ __forceinline__ void lane_src(int wave, int lane, int hidden, int& off_a, int& off_b) {
    const int r0 = wave * 8 + (lane >> 3);
    const int ch0 = (lane & 7) ^ ((r0 >> 1) & 7);
    const int urow = 32 * (r0 >> 5) + 8 * ((r0 >> 2) & 3) + 4 * ((r0 >> 4) & 1) + (r0 & 3);
    off_a = (urow * D + ch0 * 8) * 2;
    off_b = (urow * hidden + ch0 * 8) * 2;
}

// one thread per 16-byte chunk of the packed image
__global__ void mlp_pack_kernel(const char* __restrict__ w1, const char* __restrict__ w2, int hidden, char* __restrict__ out) {
    const int nchunk = hidden / 128, upt = 4 * nchunk;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // chunk index
    if (i >= (int64_t)upt * (UNIT / 16)) return;
    const int pos = (int)(i / (UNIT / 16)), o = (int)(i % (UNIT / 16)) * 16;  // byte offset inside the unit
    const int j = o / SLAB, pw = (o % SLAB) / 1024, q = pw >> 2, wave = pw & 3, lane = (o % 1024) / 16;
    const UnitSrc u = unit_src(pos, nchunk, hidden);
    int la, lb;
    lane_src(wave, lane, hidden, la, lb);
    const int src = u.base + j * u.joff + (q & 1) * u.q1 + (q >> 1) * u.q2 + (u.is_a ? la : lb);
    *(u32x4*)(out + (int64_t)pos * UNIT + o) = *(const u32x4*)((u.is_a ? w1 : w2) + src);
}

// DBG (tools/mlp_probe.hip only): 1 = no weight DMA / ring syncs, 2 = GELU replaced by a plain pack,
// 4 = no LDS fragment reads / MFMAs.
// PACKED: the weights come from the pre-packed image p.wpk (a DMA piece = 1 KiB of consecutive bytes) instead of W1 / W2.
// IMG / XIN: fragment-blocked activation images (kernels.h, "activation images") -- IMG: y1 is read and x / xn_out are
// written as images; XIN: x is read as an image.  Every load / store instruction of a row phase then touches 1 KiB of
// consecutive bytes (a row-major matrix gives 16 rows x 64 B per instruction: 35 GB/s per CU instead of 50-120).
template <int DBG = 0, bool PACKED = false, bool IMG = false, bool XIN = false>
__global__ __launch_bounds__(256, 1) void mlp_pipe_kernel(const MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gam = (float*)(smem + 3 * UNIT);
    float* bet = gam + D;
    float* b2s = bet + D;
    float* b1s = b2s + D;                  // [hidden]
    int* tile_s = (int*)(b1s + p.hidden);  // [2] tile handed to this workgroup, double-buffered by parity
    float* gam1 = (float*)(tile_s + 4);    // next block's LayerNorm-1 (gamma | beta), if p.xn_out

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int nchunk = p.hidden / 128;
    const int upt = 4 * nchunk;  // ring units (phases) per tile pass

    // ---- weight DMA: LDS row R of a slab is filled by (wave, piece q) with R = wave*8 + (lane>>3) + 32q; the
    //      16-byte chunk a lane fetches is its LDS chunk position XOR ((R>>1)&7) (same for every q) ----
    const bf16_t* W1 = (const bf16_t*)p.w1;
    const bf16_t* W2 = (const bf16_t*)p.w2;
    // A slab: 64 hidden x 128 k, LDS row R = 64*khalf + rho.  MFMA row rho = 16nf + 4g' + e of the half holds hidden
    // unit U(rho) = 32(nf>>1) + 8g' + 4(nf&1) + e: lane g' then owns, over the fragment pair (2f, 2f+1), hidden
    // 32f + 8g' + (0..7).  U(rho + 32) = U(rho) + 32, so piece q still only adds a uniform offset.
    // per-lane BYTE offsets into W1 (A slabs) and W2 (B slab: 128 outputs x 64 hidden, LDS row R = output U(R)), both multiples
    // of 16 and < 2^20: packed into ONE register (two loop-invariant registers were being spilled and re-loaded from
    // scratch, with a vmcnt(0), once per phase); everything else about a DMA address is wave-uniform.
    // With a pre-packed image (p.wpk) a piece is 1 KiB of consecutive bytes: the lane part is wave * 1024 + lane * 16.
    int la_, lb_;
    lane_src(wave, lane, p.hidden, la_, lb_);
    constexpr bool packed = PACKED;
    const uint32_t lanepack = packed ? (uint32_t)(wave * 64 + lane) : (((uint32_t)la_ >> 4) | (((uint32_t)lb_ >> 4) << 16));
    // issue side of the ring: the unit whose 12 pieces per wave are being issued
    // buffer-addressed LDS-DMA: resource = whole matrix (scalar registers), per-lane offset in ONE 32-bit register,
    // everything else (unit, slab, piece) in the scalar offset -> no vector address arithmetic per piece
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(packed ? (void*)p.wpk : (void*)W1, 0, (packed ? 2 : 1) * p.hidden * D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(packed ? (void*)p.wpk : (void*)W2, 0, (packed ? 2 : 1) * p.hidden * D * 2, 0x00020000);
    bool ia = true;        // the unit being issued is an fc1 unit
    uint32_t ilane = 0;    // per-lane byte offset
    int ioff = 0, ijoff = 0, iq1 = 0, iq2 = 0;  // bytes, wave-uniform
    int islot = 0, ipos = 0;
    auto set_issue = [&](int pos, int slot) {
        if constexpr (packed) {  // unit pos of the image, laid out exactly like its ring slot
            ia = true;
            ioff = pos * UNIT;
            ilane = lanepack << 4;
            ijoff = SLAB;
            iq1 = 4096;
            iq2 = 8192;
        } else {
            const UnitSrc u = unit_src(pos, nchunk, p.hidden);
            ia = u.is_a;
            ioff = u.base;
            ijoff = u.joff;
            iq1 = u.q1;
            iq2 = u.q2;
            ilane = (u.is_a ? (lanepack & 0xffffu) : (lanepack >> 16)) << 4;
        }
        islot = slot;
    };
    auto dma_piece = [&](auto T_) __attribute__((always_inline)) {
        constexpr int t = decltype(T_)::value, j = t >> 2, q = t & 3;
        if constexpr ((DBG & 1) == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ia ? rs1 : rs2, (LDS_AS void*)(smem + islot * UNIT + j * SLAB + (q * 4 + wave) * 1024), 16, ilane,
                                                     ioff + j * ijoff + (q & 1) * iq1 + (q >> 1) * iq2, 0, 0);
    };

    for (int i = tid; i < D; i += 256) {
        gam[i] = p.ln_w[i];
        bet[i] = p.ln_b[i];
        b2s[i] = p.b2[(i & ~31) + 8 * ((i >> 2) & 3) + 4 * ((i >> 4) & 1) + (i & 3)];  // permuted like the fc2 weight rows
        if (p.xn_out) {
            gam1[i] = p.ln_next_w[i];
            gam1[D + i] = p.ln_next_b[i];
        }
    }
    for (int i = tid; i < p.hidden; i += 256) {  // same permutation inside every 64-half: b1s[base + rho] = b1[base + U(rho)]
        const int rho = i & 63;
        b1s[i] = p.b1[(i & ~63) + 32 * (rho >> 5) + 8 * ((rho >> 2) & 3) + 4 * ((rho >> 4) & 1) + (rho & 3)];
    }
    if (tid == 0) tile_s[0] = atomicAdd(p.counter, 1);
    __syncthreads();
    int tile = __builtin_amdgcn_readfirstlane(tile_s[0]);
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memrealtime();

    // fragment read offsets inside a slab (128-byte LDS rows, chunk XOR ((row>>1)&7))
    uint32_t foff[2];
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1) foff[k1] = li * 128 + (((g + 4 * k1) ^ ((lane >> 1) & 7)) << 4);
    const uint32_t lbase = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
    const uint32_t b1base = (uint32_t)(uintptr_t)(LDS_AS char*)b1s + 16 * g;
    const uint32_t tsbase = (uint32_t)(uintptr_t)(LDS_AS char*)tile_s;
    const uint32_t gbase = (uint32_t)(uintptr_t)(LDS_AS char*)gam + 32 * g;
    const uint32_t b2base = (uint32_t)(uintptr_t)(LDS_AS char*)b2s + 16 * g;
    const uint32_t g1base = (uint32_t)(uintptr_t)(LDS_AS char*)gam1 + 32 * g;

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

    // fragment registers: two sets of 4 (one 8-MFMA group each).  Group 0 of BOTH kinds reads slab 0, rows
    // 16n + li (n = 0..3), chunk g: the cross-phase prefetch does not need to know what the next phase is.
    u32x4 wA[2][4];
    auto rd_frag = [&](auto KIND_, auto SET_, auto G_, uint32_t sa) __attribute__((always_inline)) {
        constexpr int kind = decltype(KIND_)::value, set = decltype(SET_)::value, gg = decltype(G_)::value, j = gg >> 2;
        // fc1: k-step ks = gg & 3 of slab j -> LDS rows 64(ks>>1) + 16n + li, chunk 4(ks&1) + g
        // fc2: k-step fl = (gg>>1)&1, output fragments 4(gg&1) + n -> rows 16(4(gg&1) + n) + li, chunk 4fl + g
        constexpr int k1 = kind == KA ? (gg & 1) : ((gg >> 1) & 1);
        constexpr int off = j * SLAB + (kind == KA ? ((gg & 3) >> 1) * 8192 : (gg & 1) * 8192);
        if constexpr ((DBG & 4) == 0) {
            const uint32_t a = sa + foff[k1];
            // (asm operands do not trigger the implicit capture in a generic lambda: bind references first)
            u32x4 &d0 = wA[set][0], &d1 = wA[set][1], &d2 = wA[set][2], &d3 = wA[set][3];
            DSR128(d0, a, off);
            DSR128(d1, a, off + 2048);
            DSR128(d2, a, off + 4096);
            DSR128(d3, a, off + 6144);
        }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;

    f32x4 bq[4];  // fc1 bias a half starts from: b1s[off + 16 nf + 4g + e], read one phase ahead
    auto bias_rd = [&](int off) __attribute__((always_inline)) {  // 4 reads, no wait: covered by the next counted wait
        const uint32_t a = b1base + off * 4;
        f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
        DSR128(q0, a, 0);
        DSR128(q1, a, 64);
        DSR128(q2, a, 128);
        DSR128(q3, a, 192);
    };
    // first fragments and bias of the pass (they stay in registers across a tile's row phases)

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
#ifdef MLP_WAIT_STAMP
        unsigned long long wait_vm = 0, wait_bar = 0;
#endif
        if (tid == 0) {  // next tile: fetched now, read after this tile's ring barriers
            const int nt = atomicAdd(p.counter, 1);
            asm volatile("ds_write_b32 %0, %1" ::"v"(tsbase + 4 * ((seq + 1) & 1)), "v"(nt) : "memory");
        }

        // ---- activations: v = x + y1 -> LN2 -> operand fragments ----
        u32x4 af[2][NCH];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
            int r = (wave * 2 + mf) * 16 + li;
            r = r < nrows ? r : (nrows > 0 ? nrows - 1 : 0);
            // image forms: whole fragments only (the launcher guarantees M % 16 == 0); a fragment past the tile's end
            // re-reads fragment 0 of the tile (never stored)
            const int fr = (wave * 2 + mf) * 16 < nrows ? (wave * 2 + mf) * 16 : 0;
            // x: row-major: row r, floats (g + 4c) * 8 + 4h;  image: fragment base + c * 512 + h * 256 + lane * 4
            const float* xr = XIN ? p.x + (int64_t)(row0 + fr) * D + lane * 4 : p.x + (int64_t)(row0 + r) * D + g * 8;
            constexpr int xc_ = XIN ? 512 : 32, xh_ = XIN ? 256 : 4;
            f32x4 v[NCH][2];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                v[c][0] = NT_LD((const f32x4*)(xr + c * xc_));
                v[c][1] = NT_LD((const f32x4*)(xr + c * xc_ + xh_));
            }
            if (p.y1) {
                // y1 (bf16): row-major: row r, elements (g + 4c) * 8;  image: fragment base + c * 512 + lane * 8
                const bf16_t* yr = IMG ? (const bf16_t*)p.y1 + (int64_t)(row0 + fr) * D + lane * 8 : (const bf16_t*)p.y1 + (int64_t)(row0 + r) * D + g * 8;
                constexpr int yc_ = IMG ? 512 : 32;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bf16x8 y = __builtin_bit_cast(bf16x8, NT_LD((const u32x4*)(yr + c * yc_)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[c][0][e] += (float)y[e];
                        v[c][1][e] += (float)y[4 + e];
                    }
                }
            }
            ln_rows_lds<D, NCH>(v, gbase, p.ln_eps, af[mf]);
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
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            u32x4& a2 = af[0][c];
            asm volatile("" : "+v"(a2));
        }
        PSTAMP(2);
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memtime();

        f32x4 acc2[2][NF2];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) acc2[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 acc1[2][8];
        u32x4 hf[2][4];
#pragma unroll
        for (int mf = 0; mf < 2; ++mf)
#pragma unroll
            for (int f = 0; f < 4; ++f) hf[mf][f] = u32x4{0u, 0u, 0u, 0u};
        // one 2-element GELU: unit u of half-chunk GH -> one 32-bit word of the fc2 operand fragments
        // (fragment f of fc2 takes fc1 fragments 2f (K slots 0-3) and 2f+1 (4-7): hidden 32f + 8g + slot)
        auto gelu_unit = [&](auto GH_, auto U_) __attribute__((always_inline)) {
            constexpr int gh = decltype(GH_)::value, u = decltype(U_)::value;
            constexpr int mf = u >> 3, nfl = (u >> 1) & 3, jp = u & 1, nf = 4 * gh + nfl;
            f32x2 v = {acc1[mf][nf][2 * jp], acc1[mf][nf][2 * jp + 1]};
            if constexpr ((DBG & 2) == 0) v = gelu2(v);
            hf[mf][nf >> 1][2 * (nf & 1) + jp] = pack_bf16x2(v[0], v[1]);
        };

        // ---- one phase: 12 groups of 8 MFMAs on the unit in slot cons % 3 ----
        // KIND/H: fc1 half H (into acc1[.][4H..4H+3], started from the bias in bq) or fc2 half H (hf[.][2H..2H+1])
        // GH/GSEC: GELU units of half-chunk GH, first (0) or second (1) eight, one per group 4..11; GH = -1: none
        // NB/nb: the NEXT phase is an fc1 phase and starts from the bias at b1s offset nb (read with the cross-phase
        //     prefetch).  An asm read whose result is never used must not be issued: its destination would be
        //     re-used by the compiler while the data is still landing.
        //     NB = -1: last phase of the tile, nothing is prefetched (the row phases in between need the registers)
        auto phase = [&](auto KIND_, auto H_, auto GH_, auto GSEC_, auto NB_, int nb) __attribute__((always_inline)) {
            constexpr int kind = decltype(KIND_)::value, h = decltype(H_)::value, gh = decltype(GH_)::value;
            constexpr int gsec = decltype(GSEC_)::value, needb = decltype(NB_)::value;
            const uint32_t sa = lbase + (cons % 3) * UNIT;
            const uint32_t sn = lbase + ((cons + 1) % 3) * UNIT;
            sfor<0, 12>([&](auto G_) __attribute__((always_inline)) {
                constexpr int gg = decltype(G_)::value, set = gg & 1;
                typedef std::integral_constant<int, set ^ 1> NS;
                // (1) fragment reads one group ahead
                if constexpr (gg == 11) {
                    if constexpr ((DBG & 1) == 0) {
#ifdef MLP_WAIT_STAMP  // (probe builds) cycles wave 0 spends in the two waits of a tile's 48 ring syncs -> stamp slots 12 / 13
                        const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        const unsigned long long w1_ = __builtin_amdgcn_s_memtime();
                        __builtin_amdgcn_s_barrier();
                        const unsigned long long w2_ = __builtin_amdgcn_s_memtime();
                        wait_vm += w1_ - w0_;
                        wait_bar += w2_ - w1_;
#else
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of the next unit have landed
                        __builtin_amdgcn_s_barrier();                     // ... everyone's; unit cons-1 is no longer read
#endif
                        set_issue(ipos, (cons + 2) % 3);
                        ipos = ipos + 1 == upt ? 0 : ipos + 1;
                    }
                    if constexpr (needb < 0) {
                        LGKM(0);
                    } else {
                        rd_frag(I0{}, NS{}, I0{}, sn);
                        if constexpr (needb > 0) {
                            bias_rd(nb);
                            LGKM(8);
                        } else {
                            LGKM(4);
                        }
                    }
                } else {
                    rd_frag(KIND_, NS{}, std::integral_constant<int, gg + 1>{}, sa);
                    LGKM(4);
                }
                // (2) 8 MFMAs, with the vector work that hides under them
                if constexpr ((DBG & 4) == 0) {
                    if constexpr (kind == KA) {
                        constexpr int j = gg >> 2, ks = gg & 3, kidx = 4 * j + ks;
                        if constexpr (gg == 0) {
                            // the bias read a phase ago has landed only NOW (the wait above): re-define it here, so that
                            // no copy of it (hipcc moves it to the accumulator file) can be placed before this point
                            f32x4 &q0 = bq[0], &q1 = bq[1], &q2 = bq[2], &q3 = bq[3];
                            asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
                        }
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                            for (int mf = 0; mf < 2; ++mf) {
                                if constexpr (gg == 0) {  // C operand = bias: acc1[e] = b1[.. 16nf + 4g + e] + ...
                                    f32x4 t = bq[nf];
                                    Tr<bf16_t>::mma16(t, wA[set][nf], af[mf][kidx]);
                                    acc1[mf][4 * h + nf] = t;
                                } else {
                                    Tr<bf16_t>::mma16(acc1[mf][4 * h + nf], wA[set][nf], af[mf][kidx]);
                                }
                            }
                    } else {
                        constexpr int j = gg >> 2, fl = (gg >> 1) & 1, q0 = (gg & 1) * 4;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int mf = 0; mf < 2; ++mf) Tr<bf16_t>::mma16(acc2[mf][8 * j + q0 + jj], wA[set][jj], hf[mf][2 * h + fl]);
                    }
                }
                // (4 + 4 + 4 or 3 x 4 pieces per group instead of 2 x 6: measured, no difference)
                if constexpr (gg == 11) {
                    dma_piece(std::integral_constant<int, 0>{});
                    dma_piece(std::integral_constant<int, 1>{});
                } else if constexpr (gg <= 4) {
                    dma_piece(std::integral_constant<int, 2 + 2 * gg>{});
                    dma_piece(std::integral_constant<int, 3 + 2 * gg>{});
                }
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

        // first fragments and bias of the pass (asm reads land asynchronously: nothing but the first phase may sit
        // between them and their counted wait -- in particular not the row phases, where the compiler moves registers)
        rd_frag(I0{}, I0{}, I0{}, lbase + (cons % 3) * UNIT);
        bias_rd(0);
        // chunk 0 (peeled: no runtime branches around phases inside the steady-state loop).  Its half-0 GELUs have
        // only A1(0) to hide in: the second eight run bare.
        phase(TA{}, I0{}, IM1{}, I0{}, I1{}, 64);
        phase(TA{}, I1{}, I0{}, I0{}, I0{}, 0);
        sfor<8, 16>([&](auto U_) __attribute__((always_inline)) { gelu_unit(I0{}, U_); });
        __builtin_amdgcn_sched_barrier(0);
        phase(TB{}, I0{}, I1{}, I0{}, I1{}, 128);
        PSTAMP(5);
        for (int c = 1; c < nchunk; ++c) {
            phase(TA{}, I0{}, I1{}, I1{}, I0{}, 0);             // A0(c)   + second eight GELUs of half 1 of chunk c-1
            phase(TB{}, I1{}, I0{}, I0{}, I1{}, c * 128 + 64);  // B1(c-1) + first eight of half 0 of chunk c
            phase(TA{}, I1{}, I0{}, I1{}, I0{}, 0);             // A1(c)   + second eight of half 0
            phase(TB{}, I0{}, I1{}, I0{}, I1{}, c + 1 < nchunk ? (c + 1) * 128 : 0);  // B0(c) + first eight of half 1
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
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) p.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime();
#ifdef MLP_WAIT_STAMP
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0 && seq == PSTAMP_SEQ) {
            p.stamps[(size_t)blockIdx.x * 16 + 12] = wait_vm;
            p.stamps[(size_t)blockIdx.x * 16 + 13] = wait_bar;
        }
#endif

        // ---- epilogue: x <- x + y1 + acc2 + b2 (this workgroup owns its rows: in place, no other reader).
        //      acc2[mf][nf][e] is output column 16(nf & ~1) + 8g + 4(nf & 1) + e of row li (permuted fc2 rows): fragment
        //      pair pr = nf >> 1 holds the 8 consecutive columns 32 pr + 8g + (0..7).
        //      All row loads of a fragment are issued before the first asm statement (nothing moves across those).
        sfor<0, 2>([&](auto MF_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
            constexpr int mf = decltype(MF_)::value;
            const int r = (wave * 2 + mf) * 16 + li;
            const bool live = r < nrows;
            // fragment pair pr of this lane = the 8 columns 32 pr + 8g ..: chunk (g + 4 pr) of the row -- the prologue's
            // addressing with c = pr (x: h = nf & 1).  xl: old values (layout XIN), xs: new ones (layout IMG)
            const int fr = (wave * 2 + mf) * 16;  // (image forms: stored only when live, i.e. fr < nrows)
            const float* xl = XIN ? p.x + (int64_t)(row0 + (live ? fr : 0)) * D + lane * 4 : p.x + (int64_t)(row0 + (live ? r : 0)) * D + 8 * g;
            float* xs = IMG ? p.x + (int64_t)(row0 + (live ? fr : 0)) * D + lane * 4 : p.x + (int64_t)(row0 + (live ? r : 0)) * D + 8 * g;
            const bf16_t* yr = IMG ? (const bf16_t*)p.y1 + (int64_t)(row0 + (live ? fr : 0)) * D + lane * 8
                                   : (const bf16_t*)p.y1 + (int64_t)(row0 + (live ? r : 0)) * D + 8 * g;
            constexpr int xlp_ = XIN ? 512 : 32, xlh_ = XIN ? 256 : 4, xsp_ = IMG ? 512 : 32, xsh_ = IMG ? 256 : 4, yp_ = IMG ? 512 : 32;
            f32x4 xv[NF2];  // the row's old values, then (in place) its new ones
            u32x4 yv[NF2 / 2];
#pragma unroll
            for (int nf = 0; nf < NF2; ++nf) xv[nf] = NT_LD((const f32x4*)(xl + xlp_ * (nf >> 1) + xlh_ * (nf & 1)));
            if (p.y1) {
#pragma unroll
                for (int pr = 0; pr < NF2 / 2; ++pr) yv[pr] = NT_LD((const u32x4*)(yr + yp_ * pr));
            } else {
#pragma unroll
                for (int pr = 0; pr < NF2 / 2; ++pr) yv[pr] = u32x4{0u, 0u, 0u, 0u};
            }
            // converting in place (row-major in, image out): a lane's stores land where OTHER lanes' loads read -- every
            // load of the fragment must have returned before the first store leaves
            if constexpr (XIN != IMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float rs = 0.f;
            sfor<0, NF2 / 4>([&](auto Q_) __attribute__((always_inline)) {  // 4 output fragments at a time
                constexpr int q4 = decltype(Q_)::value;
                f32x4 bb[4];
                const uint32_t ba = b2base;
                f32x4 &r0_ = bb[0], &r1_ = bb[1], &r2_ = bb[2], &r3_ = bb[3];
                DSR128X4_WAIT(r0_, r1_, r2_, r3_, ba, (4 * q4) * 64, (4 * q4 + 1) * 64, (4 * q4 + 2) * 64, (4 * q4 + 3) * 64);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nf = 4 * q4 + j;
                    f32x4 v = acc2[mf][nf] + bb[j] + xv[nf];
                    const bf16x8 y = __builtin_bit_cast(bf16x8, yv[nf >> 1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)y[4 * (nf & 1) + e];
                    if (live) NT_ST(v, (f32x4*)(xs + xsp_ * (nf >> 1) + xsh_ * (nf & 1)));
                    xv[nf] = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) rs += v[e];
                }
            });
            if (p.xn_out) {
                // LayerNorm-1 of the next block on the finished row (the 4 g-lanes of a row hold all of it), as bf16
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                const float mean = rs * (1.0f / D);
                float q = 0.f;
#pragma unroll
                for (int nf = 0; nf < NF2; ++nf)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = xv[nf][e] - mean;
                        q = __builtin_fmaf(a, a, q);
                    }
                q += __shfl_xor(q, 16, 64);
                q += __shfl_xor(q, 32, 64);
                const float rstd = 1.0f / sqrtf(q * (1.0f / D) + p.ln_eps);
                bf16_t* nr = IMG ? (bf16_t*)p.xn_out + (int64_t)(row0 + (live ? fr : 0)) * D + lane * 8
                                 : (bf16_t*)p.xn_out + (int64_t)(row0 + (live ? r : 0)) * D + 8 * g;
                sfor<0, NF2 / 2>([&](auto P_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
                    constexpr int pr = decltype(P_)::value;
                    f32x4 g0, g1, b0, b1;
                    const uint32_t ga = g1base;
                    DSR128X4_WAIT(g0, g1, b0, b1, ga, pr * 128, pr * 128 + 16, pr * 128 + D * 4, pr * 128 + D * 4 + 16);
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[e] = __builtin_fmaf((xv[2 * pr][e] - mean) * rstd, g0[e], b0[e]);
                        y[4 + e] = __builtin_fmaf((xv[2 * pr + 1][e] - mean) * rstd, g1[e], b1[e]);
                    }
                    u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(y[2 * e], y[2 * e + 1]);
                    if (live) NT_ST(o, (u32x4*)(nr + yp_ * pr));
                });
            }
        });
        PSTAMP(4);
        if (DBG & 1) __syncthreads();  // (no ring barriers in this debug build)
        int nt;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nt) : "v"(tsbase + 4 * ((seq + 1) & 1)) : "memory");
        tile = __builtin_amdgcn_readfirstlane(nt);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: pieces of a pass that never runs)
    if (HIPT_STAMPS_ON(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace

bool hipt_mlp_pipe_supported(int dtype, int D_, int hidden) {
    return dtype == HIPT_BF16 && D_ == 384 && hidden % 128 == 0 && hidden >= 256 && hidden <= 1536;
}

int hipt_mlp_pack_launch(const void* w1, const void* w2, int D_, int hidden, void* packed, hipStream_t st) {
    if (!hipt_mlp_pipe_supported(HIPT_BF16, D_, hidden)) {
        hipt_set_error("mlp pack: unsupported D=%d hidden=%d", D_, hidden);
        return HIPT_E_UNSUPPORTED;
    }
    const int64_t chunks = (int64_t)(hidden / 128) * 4 * (UNIT / 16);
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, (const char*)w1, (const char*)w2, hidden, (char*)packed);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

template <int DBG>
int hipt_mlp_pipe_launch_dbg(const MlpParams& p_in, hipStream_t st) {
    MlpParams p = p_in;
    const int lds = 3 * UNIT + (3 * D + p.hidden) * 4 + 16 + 2 * D * 4;
    // p.img: bit 0 = y1 in / x out / xn out are activation images, bit 1 = x in is one (images need the packed weights too:
    // one instantiation less; and whole 16-row fragments)
    if (p.img && (!p.wpk || (p.img & 2 && !(p.img & 1)) || p.M % 16 != 0)) {
        hipt_set_error("mlp_pipe: activation images need packed weights, M %% 16 == 0 and img in {0, 1, 3} (img=%d, M=%d)", p.img, p.M);
        return HIPT_E_BADARG;
    }
    auto k = p.img == 3 ? mlp_pipe_kernel<DBG, true, true, true>
           : p.img == 1 ? mlp_pipe_kernel<DBG, true, true, false>
           : p.wpk ? mlp_pipe_kernel<DBG, true> : mlp_pipe_kernel<DBG, false>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)mlp_pipe_kernel<DBG, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp_pipe_kernel<DBG, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp_pipe_kernel<DBG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)mlp_pipe_kernel<DBG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(mlp_pipe) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("mlp_pipe: cannot query the device");
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
    const int tail_tiles = (tiles > ncu && rem > 0 && rem <= ncu / 8) ? rem : 0;
    p.full_tiles = tiles - tail_tiles;
    const int tail_rows = p.M - p.full_tiles * TMR;
    p.ntiles = p.full_tiles + (tail_rows > 0 ? (tail_rows + 15) / 16 : 0);
    const int grid = p.ntiles < ncu ? p.ntiles : ncu;
    p.stagger = 0;
    if (hipMemsetAsync(p.counter, 0, sizeof(int), st) != hipSuccess) {
        hipt_set_error("mlp_pipe: hipMemsetAsync(counter) failed");
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
        double pro = 0, chunks = 0, epi = 0, ghz = 0;
#ifdef MLP_WAIT_STAMP
        double wvm = 0, wbar = 0;
#endif
        for (int b = 0; b < grid; ++b) {
            pro += (double)(h[b * 16 + 2] - h[b * 16 + 0]) * 0.01 / grid;
            chunks += (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.01 / grid;
            epi += (double)(h[b * 16 + 4] - h[b * 16 + 3]) * 0.01 / grid;
            ghz += (double)(h[b * 16 + 9] - h[b * 16 + 8]) / (double)(h[b * 16 + 3] - h[b * 16 + 2]) * 0.1 / grid;
#ifdef MLP_WAIT_STAMP
            wvm += (double)h[b * 16 + 12] / grid;
            wbar += (double)h[b * 16 + 13] / grid;
#endif
        }
        fprintf(stderr, "[mlp_pipe dbg=%d hidden=%d grid=%d tiles=%d(+%d)] total %.1f us | first tiles: rows+LN %.1f, chunks %.1f (%.2f GHz), epilogue %.1f\n",
                DBG, p.hidden, grid, p.full_tiles, p.ntiles - p.full_tiles, (double)(t4 - t0) * 0.01, pro, chunks, ghz, epi);
#ifdef MLP_WAIT_STAMP
        fprintf(stderr, "   ring syncs of that tile, wave 0: %.0f cycles waiting for its DMA pieces, %.0f in the barrier\n", wvm, wbar);
#endif
    }
#endif
    return HIPT_OK;
}

int hipt_mlp_pipe_launch(const MlpParams& p, hipStream_t st) { return hipt_mlp_pipe_launch_dbg<0>(p, st); }
