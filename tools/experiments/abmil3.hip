// CLAM_SB / ABMIL gated-attention pooling, bf16 [384, 128, 64], third form: WEIGHTS IN REGISTERS, BAG THROUGH AN LDS-DMA RING
// (models/model_clam.py:41-64, 83-92, 147-183; same math as abmil2.hip / abmil.hip).
//
// Why another form.  abmil2's waves stream their rows HBM -> VGPR and keep both weight matrices (128 KiB) in LDS: one wave per
// SIMD, every step's MFMAs, gate arithmetic and row loads issue serially from that one wave (in-kernel stamps: 5.1 us until
// the weight images are in LDS, then 5.5 us per 32-row step against 3.5 us of HBM time; MFMA pipe 10 % busy).  Here the roles
// are swapped:
//   * the bag streams HBM -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB pieces of consecutive bytes, the bank swizzle applied
//     on the SOURCE address) into a ring of five 32-row tiles, requested BEFORE anything else in the kernel and then three
//     tiles (72 KiB per CU) ahead of the compute: decoupled from the waves' instruction streams, no register cost; rows past a
//     workgroup's range are out of the buffer's bounds: they cost no traffic and read as zero;
//   * 8 waves per workgroup, two per SIMD (while one waits on LDS or runs the gate's transcendentals the other issues MFMAs).
//     Wave w owns hidden units [16w, 16w+16) of h1 and gate pairs [8w, 8w+8): its slice of W1 (12 KiB) and of [Wa;Wb] (4 KiB)
//     lives in REGISTERS as MFMA A-operand fragments for the whole kernel, loaded from a fragment-ordered image (1 KiB of
//     consecutive bytes per load instruction) -- no weight image in LDS;
//   * software pipeline over 32-row tiles with ONE barrier per tile.  Between barrier t and barrier t + 1 a wave runs three
//     independent pieces of work, each on data the barrier has just made complete: the logits / running max / pooling of
//     tile t - 1 (sums the 8 waves' partial logits in a fixed order: a row's logit does not depend on where the row sits;
//     pools ITS 16 columns of the un-rounded fp32 h1), phase 2 of tile t (8 MFMAs: the whole bf16 h1 row from the LDS
//     exchange image as B operand against the wave's gate slice -> tanh * sigmoid -> per-wave partial logits to LDS) and
//     phase 1 of tile t + 1 (24 MFMAs: x from the ring as B operand -> ReLU -> its bf16 h1 slice to the other exchange
//     image).  No cross-wave reduction at the end: wave w's pool is columns [16w, 16w+16) of the workgroup's partial.
//   * partial (max, sum, acc[128]) per workgroup; the last arriver (self-resetting ticket) merges and classifies.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int S0 = 384, S1 = 128, S2 = 64;
constexpr int NW = 4;                        // waves per workgroup, one per SIMD (the whole 512-register file each)
constexpr int SROWS = 32;                    // rows per step (2 MFMA row fragments) = one ring tile = one barrier
constexpr int ROWB = S0 * 2;                 // 768 bytes per bag row
constexpr int TILEB = 32 * ROWB;             // 24 KiB ring slot = 32 rows = 24 DMA pieces, 6 per wave
constexpr int NSLOT = 5;                     // ring depth in 32-row tiles: 3 (72 KiB) in flight beside the super tile being read
constexpr int H1B = SROWS * S1 * 2;          // 8 KiB: bf16 h1 exchange image, 256-byte rows (two of them: steps alternate)
constexpr int AXB = SROWS * NW * 4;          // 1 KiB: per-wave partial logits [row][wave] (two of them)
constexpr int LDS_BYTES = NSLOT * TILEB + 2 * H1B + 2 * AXB + 256;
constexpr int PK_WAVE = 32 * 1024;           // packed weight image: 24 KiB of W1 + 8 KiB of [Wa;Wb] per wave
constexpr float LOG2E = 1.4426950408889634f;

#define DSR128I(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// the same into the ACCUMULATOR file ("a"): the x images (2 x 96 registers) live there -- an MFMA takes its B operand from either
// file -- so that hipcc has no reason to copy a read's destination anywhere before the counted wait that retires it
#define DSR128A(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(dst) : "v"(addr), "n"(off))
#define DSW64I(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define DSW32I(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define LGKM3(n)                                                \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
    __builtin_amdgcn_sched_barrier(0)
#define VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int I, int N, class F> __device__ __forceinline__ void sfor3(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor3<I + 1, N>(f);
    }
}

// tanh(x) sigmoid(y) w for two gate pairs at once; v = (a_j, a_j+1, b_j, b_j+1) with the biases in, cw = (wc_j, wc_j+1)
// (contraction off: a row's logit must not depend on the fragment / tile it lands in)
__device__ __forceinline__ f32x2 gate_pair3(const f32x4& v, const f32x2& cw) {
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

__device__ __forceinline__ f32x4 ld4sc1(__amdgpu_buffer_rsrc_t r, int off) {  // 16 bytes at an 8-byte aligned offset, sc1
    const u32x2 lo = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 16), hi = __builtin_amdgcn_raw_buffer_load_b64(r, off + 8, 0, 16);
    const f32x2 l = __builtin_bit_cast(f32x2, lo), h = __builtin_bit_cast(f32x2, hi);
    return f32x4{l[0], l[1], h[0], h[1]};
}

// Packed weight image for the ring kernel (made once per set of weights by hipt_clam_pack_ring): per wave w (0..3)
//   [hf 0..1][c 0..11][lane][16 B]  = W1[32 w + 16 hf + li][32 c + 8 g ..+7]             (24 KiB)
//   [gf 0..1][c 0..3][lane][16 B]   = [Wa;Wb] packed row 32 w + 16 gf + li, k 32 c + 8 g    (8 KiB; packed row R: quad R >> 2 holds
//                                      (a_j, a_j+1, b_j, b_j+1), j = 2 (R >> 2), i.e. source row ((R & 3) >> 1) * 64 + (R >> 2) * 2 + (R & 1))
// so that every load instruction of a wave reads 1 KiB of consecutive bytes (the row-major fragment pattern, 16 rows x 64 B per
// instruction, reads the same 128 KiB 2.3x slower with every CU at it).
__global__ void abmil_ring_pack_kernel(const bf16_t* __restrict__ w1, const bf16_t* __restrict__ wab, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk of the image
    if (i >= NW * PK_WAVE / 16) return;
    const int w = i / (PK_WAVE / 16), r = i % (PK_WAVE / 16), lane = r & 63, li = lane & 15, g = lane >> 4;
    const int blk = r >> 6;  // 0..23: W1 (hf = blk / 12, c = blk % 12); 24..31: gate (gf = (blk - 24) / 4, c = (blk - 24) % 4)
    if (blk < 24) {
        const int hf = blk / 12, c = blk % 12;
        out[i] = *(const u32x4*)(w1 + (int64_t)(32 * w + 16 * hf + li) * S0 + 32 * c + 8 * g);
    } else {
        const int gf = (blk - 24) / 4, c = (blk - 24) % 4;
        const int R = 32 * w + 16 * gf + li;
        const int srow = ((R & 3) >> 1) * S2 + (R >> 2) * 2 + (R & 1);
        out[i] = *(const u32x4*)(wab + (int64_t)srow * S1 + 32 * c + 8 * g);
    }
}

__global__ __launch_bounds__(256, 1) void abmil_ring_kernel(const bf16_t* __restrict__ bag, int N, int rows_per_wg, const u32x4* __restrict__ wpk,
                                                             const float* __restrict__ b1, const float* __restrict__ bab,
                                                             const float* __restrict__ wc, const float* __restrict__ bc, float* __restrict__ A_raw,
                                                             float* __restrict__ partials, int attention_only, unsigned* __restrict__ ticket,
                                                             const float* __restrict__ wcls, const float* __restrict__ bcls, int C, float* __restrict__ M,
                                                             float* __restrict__ logits, float* __restrict__ Y_prob, int64_t* __restrict__ Y_hat,
                                                             unsigned long long* __restrict__ stamps, int dbg) {
#define RSTAMP(k)                                                                                                                 \
    do {                                                                                                                          \
        if (HIPT_STAMPS_ON(stamps) && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    RSTAMP(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ring [NSLOT][TILEB] | h1 [2][32][256 B] | Ax [2][32][4] f32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;

    const int rbeg = blockIdx.x * rows_per_wg;
    int nrows = N - rbeg;
    nrows = nrows < rows_per_wg ? nrows : rows_per_wg;
    nrows = nrows > 0 ? nrows : 0;
    const int nst = (nrows + SROWS - 1) / SROWS;

    // ---- ring DMA first: 32-row tile u -> slot u % NSLOT; piece p of a tile = LDS bytes [1024 p, 1024 p + 1024) of its slot.
    //      The image is row-major (768-byte rows) with the 16-byte chunks of a row XOR-swizzled inside groups of 16 by the row:
    //      the 16 rows a ds_read_b128 reads at one logical chunk sit on 16 different bank quads, and a piece still reads whole
    //      256-byte runs.  Rows past the workgroup's range are out of the buffer's bounds: no traffic, zeros. ----
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(bag + (int64_t)(nrows ? rbeg : 0) * S0), 0, nrows * ROWB, 0x00020000);
    int goff[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int byte = 1024 * (wave + NW * k) + 16 * lane;
        const int r = byte / ROWB, q = (byte - r * ROWB) >> 4;
        goff[k] = r * ROWB + (((q & ~15) | ((q & 15) ^ (r & 15))) << 4);
    }
    auto dma_tile = [&](int u) __attribute__((always_inline)) {
        LDS_AS char* slot = (LDS_AS char*)smem + (u % NSLOT) * TILEB;
        const int tb = u * TILEB;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(slot + 1024 * (wave + NW * k)), 16, tb + goff[k], 0, 0, 0);
    };
    dma_tile(0);  // the first three tiles; the other two go out once the weights are in (see below)
    dma_tile(1);
    dma_tile(2);
    RSTAMP(1);

    // ---- this wave's weight slices into registers (MFMA A-operand fragments; 1 KiB of consecutive bytes per instruction).
    //      They are requested right behind tiles 0..2; hipcc waits for them with vmcnt(0) (it does not count around LDS-DMA),
    //      which at that point covers exactly what the pipeline's first two steps need.  Tiles 3 and 4 are issued after that
    //      wait, so that it does not cover them ----
    u32x4 w1f[2][12], wgf[2][4];
    {
        const u32x4* src = wpk + (size_t)wave * (PK_WAVE / 16) + lane;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int c = 0; c < 12; ++c) w1f[hf][c] = src[(hf * 12 + c) * 64];
#pragma unroll
        for (int gf = 0; gf < 2; ++gf)
#pragma unroll
            for (int c = 0; c < 4; ++c) wgf[gf][c] = src[(24 + gf * 4 + c) * 64];
    }
    f32x4 b1v[2], gbv[2];
    f32x2 cwv[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) b1v[hf] = *(const f32x4*)(b1 + 32 * wave + 16 * hf + 4 * g);  // h1 columns 32 w + 16 hf + 4 g + e
#pragma unroll
    for (int gf = 0; gf < 2; ++gf) {
        const int j0 = 16 * wave + 8 * gf + 2 * g;  // gate pairs j0, j0 + 1 of this lane in gate fragment gf
        gbv[gf] = f32x4{bab[j0], bab[j0 + 1], bab[S2 + j0], bab[S2 + j0 + 1]};
        cwv[gf] = f32x2{wc[j0], wc[j0 + 1]};
    }
    float bcv = bc[0];

    const uint32_t lbase = lds_addr(smem);
    uint32_t xoff[4];  // B-operand read of chunk q = 4 c + g of row li of a 16-row fragment: + 256 * (c >> 2)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) xoff[cc] = li * ROWB + (((4 * cc + g) ^ li) << 4);
    const uint32_t h1base = lbase + NSLOT * TILEB;
    uint32_t hoff[4];  // h1 image: row * 256 + ((chunk ^ (row & 15)) << 4)
#pragma unroll
    for (int c = 0; c < 4; ++c) hoff[c] = h1base + li * 256 + (((4 * c + g) ^ li) << 4);
    uint32_t hwr[2];   // my 4 bf16 of row li in hidden fragment hf (+ f * 4096): chunk 4 w + 2 hf + (g >> 1), half g & 1
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) hwr[hf] = h1base + li * 256 + (((4 * wave + 2 * hf + (g >> 1)) ^ li) << 4) + 8 * (g & 1);
    const uint32_t axbase = h1base + 2 * H1B;
    const uint32_t axwr = axbase + (li * NW + wave) * 4;   // Ax[row][wave] (+ f * 256, + AXB for odd tiles)
    const uint32_t axrd = axbase + li * (NW * 4);          // the 4 partials of row li (+ f * 256, + AXB for odd tiles)
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)(A_raw + (nrows ? rbeg : 0)), 0, nrows * 4, 0x00020000);

    float m_run = -INFINITY, l_lane = 0.f;
    f32x4 pool[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    // (the weights are consumed here: their wait -- everything requested so far -- happens now, outside the loop)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int c = 0; c < 12; ++c) asm volatile("" : "+v"(w1f[hf][c]));
#pragma unroll
    for (int gf = 0; gf < 2; ++gf)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(wgf[gf][c]));
    asm volatile("" : "+v"(gbv[0]), "+v"(gbv[1]), "+v"(cwv[0]), "+v"(cwv[1]), "+v"(b1v[0]), "+v"(b1v[1]), "+v"(bcv));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dma_tile(3);
    dma_tile(4);
    __builtin_amdgcn_s_barrier();  // tiles 0..2 are in LDS for everybody (raw: no __syncthreads() while ring DMA may be in flight)
    RSTAMP(2);

    // ---- the software pipeline.  All LDS traffic goes through asm (hipcc must not see LDS accesses while the ring's DMA is in
    //      flight); a tile's x operands are read into REGISTERS one step before its MFMAs (two register images X0 / X1), so the
    //      LDS reads of tile t + 2 run under the MFMAs of tiles t and t + 1 -- one wave per SIMD: nothing else would hide them.
    //      Step t, between barrier t and barrier t + 1:
    //        reads: Ax(t-1) partial logits, h1(t) exchange image, x(t+2) from the ring      (issued together, in this order)
    //        phase 2 of tile t (16 MFMAs + gate) -> Ax(t);  logits / max / pooling of tile t - 1;  phase 1 of tile t + 1 (48 MFMAs)
    //        on the x image read in step t - 1 -> h(t+1), bf16 slice to the other exchange image
    //      VM operations of a wave, in issue order: prologue D0 D1 D2 [waited] D3 D4; per step t: D(t+5) (6 pieces), S(t-1) (the
    //      A_raw store of the tile finished in the step).  x(t+3) is read behind barrier t + 1: every wave waits for ITS pieces
    //      of tile t + 3 at the end of step t: D(t+3) is followed by D(t+4), D(t+5) = 12 pieces and by the stores S(t-3)..S(t-1)
    //      that exist = min(t, 3). ----
// (lgkmcnt is a 4-bit counter: a wait can leave at most 15 LDS operations outstanding, so a tile's 24 x reads go out as two
//  batches of 12, one per row fragment)
#define RDXF(X, u, f)                                                                                      \
    do {                                                                                                   \
        const uint32_t sb_ = lbase + ((u) % NSLOT) * TILEB;                                                \
        const uint32_t a0_ = sb_ + xoff[0], a1_ = sb_ + xoff[1], a2_ = sb_ + xoff[2], a3_ = sb_ + xoff[3]; \
        DSR128A(X[f][0], a0_, (f)*12288);       DSR128A(X[f][1], a1_, (f)*12288);       DSR128A(X[f][2], a2_, (f)*12288);        DSR128A(X[f][3], a3_, (f)*12288);        \
        DSR128A(X[f][4], a0_, (f)*12288 + 256); DSR128A(X[f][5], a1_, (f)*12288 + 256); DSR128A(X[f][6], a2_, (f)*12288 + 256);  DSR128A(X[f][7], a3_, (f)*12288 + 256);  \
        DSR128A(X[f][8], a0_, (f)*12288 + 512); DSR128A(X[f][9], a1_, (f)*12288 + 512); DSR128A(X[f][10], a2_, (f)*12288 + 512); DSR128A(X[f][11], a3_, (f)*12288 + 512); \
    } while (0)
    // phase 1 of tile u from the register image X: h[f][hf] = ReLU(x W1_slice^T + b1) (lane: row 16 f + li, hidden 32 w + 16 hf
    // + 4 g + e); the bf16 slices to exchange image u & 1
#define PHASE1(X, u, h)                                                                        \
    do {                                                                                       \
        const uint32_t hw0_ = hwr[0] + ((u)&1) * H1B, hw1_ = hwr[1] + ((u)&1) * H1B;           \
        _Pragma("unroll") for (int f = 0; f < 2; ++f) {                                        \
            f32x4 acc0 = b1v[0], acc1 = b1v[1];                                                \
            _Pragma("unroll") for (int c = 0; c < 12; ++c) {                                   \
                Tr<bf16_t>::mma16(acc0, w1f[0][c], X[f][c]);                                   \
                Tr<bf16_t>::mma16(acc1, w1f[1][c], X[f][c]);                                   \
            }                                                                                  \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                    \
                acc0[e] = fmaxf(acc0[e], 0.f);                                                 \
                acc1[e] = fmaxf(acc1[e], 0.f);                                                 \
            }                                                                                  \
            h[f][0] = acc0;                                                                    \
            h[f][1] = acc1;                                                                    \
            u32x2 pk0, pk1;                                                                    \
            pk0[0] = pack_bf16x2(acc0[0], acc0[1]);                                            \
            pk0[1] = pack_bf16x2(acc0[2], acc0[3]);                                            \
            pk1[0] = pack_bf16x2(acc1[0], acc1[1]);                                            \
            pk1[1] = pack_bf16x2(acc1[2], acc1[3]);                                            \
            if (f == 0) {                                                                      \
                DSW64I(hw0_, pk0, 0);                                                          \
                DSW64I(hw1_, pk1, 0);                                                          \
            } else {                                                                           \
                DSW64I(hw0_, pk0, 4096);                                                       \
                DSW64I(hw1_, pk1, 4096);                                                       \
            }                                                                                  \
        }                                                                                      \
    } while (0)
    // one pipeline step (see above); XR: register image the reads of tile t + 2 go to, XC: image of tile t + 1 (read a step ago),
    // hP: fp32 h of tile t - 1 (finished here), hN: receives tile t + 1's
#define STEP(t, XR, XC, hP, hN)                                                                                   \
    do {                                                                                                          \
        LGKM3(0);                                                                                                 \
        __builtin_amdgcn_s_barrier();                                                                             \
        dma_tile((t) + NSLOT);                                                                                    \
        f32x4 pa0_, pa1_;                                                                                         \
        u32x4 hb_[2][4];                                                                                          \
        {                                                                                                         \
            const uint32_t ar_ = axrd + (((t) + 1) & 1) * AXB, hs_ = ((t)&1) * H1B;                               \
            const uint32_t q0_ = hoff[0] + hs_, q1_ = hoff[1] + hs_, q2_ = hoff[2] + hs_, q3_ = hoff[3] + hs_;    \
            DSR128I(pa0_, ar_, 0);                                                                                \
            DSR128I(pa1_, ar_, 256);                                                                              \
            DSR128I(hb_[0][0], q0_, 0);    DSR128I(hb_[0][1], q1_, 0);    DSR128I(hb_[0][2], q2_, 0);    DSR128I(hb_[0][3], q3_, 0);    \
            DSR128I(hb_[1][0], q0_, 4096); DSR128I(hb_[1][1], q1_, 4096); DSR128I(hb_[1][2], q2_, 4096); DSR128I(hb_[1][3], q3_, 4096); \
        }                                                                                                         \
        if ((t) + 2 < nst) {                                                                                      \
            RDXF(XR, (t) + 2, 0);                                                                                 \
            LGKM3(12);                                                                                            \
        } else {                                                                                                  \
            LGKM3(0);                                                                                             \
        }                                                                                                         \
        phase2_compute(hb_, (t));                                                                                 \
        if ((t) + 2 < nst) RDXF(XR, (t) + 2, 1);                                                                  \
        if ((t) > 0) finish_tile((t)-1, hP, pa0_, pa1_);                                                          \
        if ((t) + 1 < nst) PHASE1(XC, (t) + 1, hN);                                                               \
        if ((t) >= 3) {                                                                                           \
            VMCNT(15);                                                                                            \
        } else if ((t) == 2) {                                                                                    \
            VMCNT(14);                                                                                            \
        } else if ((t) == 1) {                                                                                    \
            VMCNT(13);                                                                                            \
        } else {                                                                                                  \
            VMCNT(12);                                                                                            \
        }                                                                                                         \
    } while (0)

    // phase 2 of tile u from the h1 fragments hb: (a | b) = h1 [Wa;Wb]_slice^T + bias; gate; partial logits to Ax image u & 1
    auto phase2_compute = [&](const u32x4 (&hb)[2][4], int u) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        f32x4 acc[2][2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            acc[f][0] = gbv[0];
            acc[f][1] = gbv[1];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                Tr<bf16_t>::mma16(acc[f][0], wgf[0][c], hb[f][c]);
                Tr<bf16_t>::mma16(acc[f][1], wgf[1][c], hb[f][c]);
            }
        float v[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const f32x2 g0 = gate_pair3(acc[f][0], cwv[0]), g1 = gate_pair3(acc[f][1], cwv[1]);
            v[f] = ((g0[0] + g0[1]) + g1[0]) + g1[1];
        }
        v[0] += __shfl_xor(v[0], 16, 64);
        v[1] += __shfl_xor(v[1], 16, 64);
        v[0] += __shfl_xor(v[0], 32, 64);
        v[1] += __shfl_xor(v[1], 32, 64);
        if (g == 0) {
            const uint32_t aw = axwr + (u & 1) * AXB;
            const float v0 = v[0], v1 = v[1];
            DSW32I(aw, v0, 0);
            DSW32I(aw, v1, 256);
        }
    };
    // logits of tile u from the 4 waves' partials (wave order), A_raw, running max, pooling of my 32 columns of h
    auto finish_tile = [&](int u, const f32x4 (&h)[2][2], const f32x4& pa0, const f32x4& pa1) __attribute__((always_inline)) {
        float a_row[2];
        {
#pragma clang fp contract(off)
            const float s0 = ((pa0[0] + pa0[1]) + pa0[2]) + pa0[3], s1 = ((pa1[0] + pa1[1]) + pa1[2]) + pa1[3];
            const int r = u * SROWS + li;
            a_row[0] = r < nrows ? s0 + bcv : -INFINITY;
            a_row[1] = r + 16 < nrows ? s1 + bcv : -INFINITY;
        }
        {   // A_raw: wave w stores rows 8 w .. 8 w + 7 of the tile.  ONE store instruction per wave and tile whatever the rows
            // (unselected lanes aim past the buffer: dropped), so that the vmcnt arithmetic of the loop holds for every tile
            const int f = wave >> 1, r = u * SROWS + f * 16 + li;
            const bool mine = g == 0 && (li >> 3) == (wave & 1) && r < nrows;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, wave < 2 ? a_row[0] : a_row[1]), ars, mine ? r * 4 : 0x7ffffff0, 0, 0);
        }
        if (!attention_only) {
            float mt = fmaxf(a_row[0], a_row[1]);
            mt = fmaxf(mt, __shfl_xor(mt, 1, 64));
            mt = fmaxf(mt, __shfl_xor(mt, 2, 64));
            mt = fmaxf(mt, __shfl_xor(mt, 4, 64));
            mt = fmaxf(mt, __shfl_xor(mt, 8, 64));  // finite: every tile has at least one valid row
            const float m_new = fmaxf(m_run, mt);
            const float resc = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);  // 0 on the first tile
            m_run = m_new;
            const float p0 = __builtin_amdgcn_exp2f((a_row[0] - m_new) * LOG2E), p1 = __builtin_amdgcn_exp2f((a_row[1] - m_new) * LOG2E);
            l_lane = l_lane * resc + p0 + p1;
            pool[0] = pool[0] * resc + h[0][0] * p0 + h[1][0] * p1;
            pool[1] = pool[1] * resc + h[0][1] * p0 + h[1][1] * p1;
        }
    };

    u32x4 X0[2][12], X1[2][12];
    f32x4 hA[2][2], hB[2][2];  // fp32 h of even / odd tiles
    if (nst > 0) {
        RDXF(X0, 0, 0);
        RDXF(X0, 0, 1);
        LGKM3(0);
        if (nst > 1) {
            RDXF(X1, 1, 0);
            RDXF(X1, 1, 1);
        }
        PHASE1(X0, 0, hA);
    }
    for (int t = 0; t < nst; t += 2) {
        STEP(t, X0, X1, hB, hB);          // reads x(t+2) -> X0; phase 1 of tile t + 1 from X1 -> hB; finishes tile t - 1 (hB) first
        if (t + 1 >= nst) break;
        STEP(t + 1, X1, X0, hA, hA);      // reads x(t+3) -> X1; phase 1 of tile t + 2 from X0 -> hA; finishes tile t (hA) first
    }
    {   // the last tile: its partial logits need one more barrier
        LGKM3(0);
        __builtin_amdgcn_s_barrier();
        if (nst > 0) {
            f32x4 pa0, pa1;
            const uint32_t ar = axrd + ((nst - 1) & 1) * AXB;
            DSR128I(pa0, ar, 0);
            DSR128I(pa1, ar, 256);
            LGKM3(0);
            if ((nst - 1) & 1) finish_tile(nst - 1, hB, pa0, pa1);
            else finish_tile(nst - 1, hA, pa0, pa1);
        }
    }
#undef STEP
#undef PHASE1
#undef RDXF
    RSTAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's last (out-of-range) pieces: nothing may still write LDS from here on
    if (attention_only) return;

    // ---- workgroup partial: every wave holds the same (max, sum); wave w holds columns [32 w, 32 w + 32) ----
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        l_lane += __shfl_xor(l_lane, o, 64);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 4; ++e) pool[hf][e] += __shfl_xor(pool[hf][e], o, 64);
    }
    float* pw = partials + (int64_t)blockIdx.x * (2 + S1);
    // (agent-scope relaxed stores = global_store sc0 sc1: they leave the XCD's L2, the merging workgroup reads them with sc1
    //  loads and no fence; hand-off table row 1 of MI355X_MICROARCH.md)
    if (tid == 0) {
        __hip_atomic_store(&pw[0], nst > 0 ? m_run : -INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&pw[1], l_lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (li == 0) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int e = 0; e < 4; ++e) __hip_atomic_store(&pw[2 + 32 * wave + 16 * hf + 4 * g + e], pool[hf][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RSTAMP(4);
    if (!ticket) return;
    // ---- fused combine (model_clam.py:180-183) by the workgroup whose ticket is the last one ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = (float*)smem;  // the ring is dead now
    int* flag = (int*)(red + 2600);
    if (tid == 0) {
        const bool last = atomicAdd(ticket, 1u) == gridDim.x - 1;
        *flag = last;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next call
    }
    __syncthreads();
    if (!*flag) return;
    {
        const int G = gridDim.x, stride = 2 + S1;
        float* Fs = red;                   // [256] rescale factors
        float* Cs = Fs + 256;              // [8][128] column partial sums
        float* Ms = Cs + 8 * S1;           // [128]
        float* Ls = Ms + S1;               // [C <= 64]
        float* wr = Ls + 64;               // [4]
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)partials, 0, G * stride * 4, 0x00020000);
        constexpr int SC1 = 16;
        // 32 threads x 16 B cover one partial row, 8 rows per pass, 32 passes = 256 rows; rows past G are out of the buffer's
        // range and read as zero.  All loads are requested before anything is reduced (one round trip).
        const int c4 = tid & 31, part = tid >> 5;
        f32x2 ml = {-INFINITY, 0.f};
        if (tid < G) {
            ml[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, tid * stride * 4, 0, SC1));
            ml[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, tid * stride * 4 + 4, 0, SC1));
        }
        f32x4 rowv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) rowv[i] = ld4sc1(prs, ((part + 8 * i) * stride + 2 + 4 * c4) * 4);
        float mx = wave_max(ml[0]);
        if (lane == 0) wr[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(wr[0], wr[1]), fmaxf(wr[2], wr[3]));
        __syncthreads();
        const float fsc = tid < G ? expf(ml[0] - mx) : 0.f;
        Fs[tid] = fsc;
        float ls = wave_sum(ml[1] * fsc);
        if (lane == 0) wr[wave] = ls;
        __syncthreads();
        const float L = (wr[0] + wr[1]) + (wr[2] + wr[3]);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 32; ++i) a += f32x4{0.f, 0.f, 0.f, 0.f} + rowv[i] * Fs[part + 8 * i];
        *(f32x4*)(Cs + part * S1 + 4 * c4) = a;
        __syncthreads();
        if (tid < S1) {
            float sacc = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) sacc += Cs[p * S1 + tid];
            sacc /= L;
            Ms[tid] = sacc;
            M[tid] = sacc;
        }
        __syncthreads();
        for (int k = wave; k < C; k += NW) {
            float sacc = Ms[lane] * wcls[(int64_t)k * S1 + lane] + Ms[lane + 64] * wcls[(int64_t)k * S1 + lane + 64];
            sacc = wave_sum(sacc);
            if (lane == 0) Ls[k] = sacc + bcls[k];
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
    RSTAMP(5);
}

}  // namespace

bool hipt_clam_ring_supported(const hipt_clam_weights* w) {
    // opt-in (HIPT_ABMIL_RING=1): on MI355X it runs the 100 000 x 384 bag in the same time as abmil2's streaming kernel (34-37 us
    // in-kernel) and covers S0 = 384 only, so the streaming kernel stays the default
    static const bool on = getenv("HIPT_ABMIL_RING") != nullptr;
    return on && w->dtype == HIPT_BF16 && w->s0 == S0 && w->s1 == S1 && w->s2 == S2 && w->ring_pk != nullptr;
}

size_t hipt_clam_ring_packed_bytes(const hipt_clam_weights* w) {
    return (w && w->dtype == HIPT_BF16 && w->s0 == S0 && w->s1 == S1 && w->s2 == S2) ? (size_t)NW * PK_WAVE : 0;
}

int hipt_clam_ring_pack_launch(const hipt_clam_weights* w, void* out, hipStream_t st) {
    hipLaunchKernelGGL(abmil_ring_pack_kernel, dim3(NW * PK_WAVE / 16 / 256), dim3(256), 0, st, (const bf16_t*)w->w1, (const bf16_t*)w->wab, (u32x4*)out);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_clam_ring_launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* partials, int* n_partials,
                          unsigned* ticket, float* M, float* logits, float* Y_prob, int64_t* Y_hat, hipStream_t st) {
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipFuncSetAttribute((const void*)abmil_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(abmil ring) failed");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    // contiguous row ranges, one workgroup per CU (at most 256 partials for the in-kernel merge)
    int grid = once.ncu[dev] < 256 ? once.ncu[dev] : 256;
    const int tiles = (N + 31) / 32;
    if (grid > tiles) grid = tiles;
    const int rows = (N + grid - 1) / grid;
    grid = (N + rows - 1) / rows;
    const bool fuse = !attention_only && ticket && M && w->n_classes <= 64;
#ifdef HIPT_DEBUG_STAMPS
    static const bool want_stamps = getenv("HIPT_ABMIL_STAMPS") != nullptr;
    static unsigned long long* dbuf = nullptr;
    if (want_stamps && !dbuf) (void)hipMalloc(&dbuf, 512 * 8 * sizeof(unsigned long long));
#else
    constexpr bool want_stamps = false;
    constexpr unsigned long long* dbuf = nullptr;
#endif
    const int lds = LDS_BYTES;
#ifdef HIPT_DEBUG_STAMPS
    static const int dbg = getenv("HIPT_ABMIL_DBG") ? atoi(getenv("HIPT_ABMIL_DBG")) : 0;  // ablations: 1 = no phase 1, 2 = no phase 2
#else
    constexpr int dbg = 0;
#endif
    hipLaunchKernelGGL(abmil_ring_kernel, dim3(grid), dim3(256), lds, st, (const bf16_t*)bag, N, rows, (const u32x4*)w->ring_pk, w->b1, w->bab, w->wc, w->bc,
                       A_raw, partials, attention_only, fuse ? ticket : nullptr, w->wcls, w->bcls, w->n_classes, M, logits, Y_prob, Y_hat,
                       want_stamps ? dbuf : nullptr, dbg);
    HIPT_CHECK_LAUNCH();
#ifdef HIPT_DEBUG_STAMPS
    if (want_stamps && grid <= 512) {
        static unsigned long long h[512 * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbuf, (size_t)grid * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t5 = 0;
        for (int b = 0; b < grid; ++b) {
            if (h[b * 8] < t0) t0 = h[b * 8];
            for (int k = 3; k < 6; ++k)
                if (h[b * 8 + k] > t5) t5 = h[b * 8 + k];
        }
        double ph[4] = {0, 0, 0, 0}, smax = 0;
        for (int b = 0; b < grid; ++b) {
            for (int k = 0; k < 4; ++k) ph[k] += (double)(h[b * 8 + k + 1] - h[b * 8 + k]) * 0.01 / grid;
            const double s0 = (double)(h[b * 8] - t0) * 0.01;
            if (s0 > smax) smax = s0;
        }
        fprintf(stderr, "[abmil ring N=%d grid=%d rows/wg=%d] total %.1f us | start<=%.1f; DMA issue %.1f; weights->regs + first tiles landed %.1f; tiles %.1f; partial %.1f\n",
                N, grid, rows, (double)(t5 - t0) * 0.01, smax, ph[0], ph[1], ph[2], ph[3]);
    }
#endif
    *n_partials = fuse ? 0 : grid;
    return HIPT_OK;
}
