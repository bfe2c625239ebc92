// CLAM_SB / ABMIL gated-attention pooling, bf16 hot configuration [S0, 128, 64] (S0 = 384: the BASELINE bag; 192: the slide
// aggregator over HIPT_4K's region features), streaming form on 32x32x16 MFMAs
// (models/model_clam.py:41-64, 83-92, 147-183; same math as abmil.hip, which remains the general kernel).
//
// HBM-bound design: the 100 000 x 384 bf16 bag (76.8 MB) is read exactly once; nothing else moves.
//   * one 4-wave workgroup per CU (one wave per SIMD, the whole register file each); the weights are staged ONCE per workgroup
//     into LDS as MFMA A-operand fragments of 1 KiB (W1: [k-step][hidden tile]; [Wa;Wb]: [k-step][gate tile], its k order
//     permuted to the accumulator-as-operand order of the h1 tiles);
//   * a WAVE owns 32-row blocks end to end (block b -> wave b mod #waves): no barrier, no exchange between waves in steady state.
//     A block's 24 KiB come HBM -> registers in WHOLE 128-byte lines (a wave instruction = 8 rows x 128 B: the lane that would
//     feed the MFMA holds row = lane, so operand-layout loads are 32 rows x 32 B per instruction, and that scatter costs a third
//     of the achievable HBM rate: 29.7 -> 24.9 us for the kernel's memory side alone); a block's worth of requests is always in
//     flight in registers, re-requested in place as soon as a slice has been used.  Slices of 4 k-steps (32 rows x 128 B) pass
//     through a 4.5 KiB LDS buffer of the wave (written in line layout, read back as B operands of v_mfma_f32_32x32x16_bf16, row
//     on the lane; a wave's LDS operations execute in order, so there is no wait between the two) one slice ahead of the MFMAs.
//     Rows past the bag read as zero through a range-checked buffer, no traffic;
//   * h1^T = W1 x^T + b1 (4 hidden tiles x KS k-steps), ReLU, packed IN PLACE as the B operand of the gate product
//     [a;b]^T = [Wa;Wb] h1^T (accumulator-as-operand: no data movement); the gate tiles hold a_j and b_j of a row in the same
//     lane, so tanh * sigmoid * wc is lane-local and a row's logit is one cross-half add;
//   * softmax pooling WITHOUT a running maximum: |A - bc| <= sum |wc| =: B because tanh * sigmoid lies in (-1, 1), so every
//     exponent is taken against the FIXED shift bc, the centre of the interval the logits can lie in: p = e^(A - bc) lies in
//     [e^-B, e^B], and for B < 60 neither p nor sum p (N < 2^22 rows) nor sum p h1 can leave fp32's range (e^88).  The caller
//     supplies B (hipt_clam_weights.logit_bound); larger / unknown bounds take the general kernel.  No rescale of the pooled
//     sums per step, and the cross-workgroup merge is a plain sum of (sum p, sum p h1[128]);
//   * biases ride in the GEMMs: a bias is split into three bf16 pieces (hi + mid + lo = the fp32 value to its last bit) that sit in
//     three k-slots of one extra k-step whose other operand is 1, 1, 1, 0, ..: the first MFMA of every accumulator chain (C = 0)
//     -- no accumulator initialisation, no bias tables;
//   * software pipeline, one wave per SIMD: the MFMA pipe of a SIMD is fed by ONE wave, so everything else that wave does sits in
//     the gaps of its own MFMA stream (24 of every 32 cycles are free for vector instructions): the gate arithmetic, the logit
//     and the pooling of block s-1 are dealt out under the 100 phase-1 MFMAs of block s, the ReLU / bf16 packing of block s under
//     its own gate-product MFMAs;
//   * the last workgroup to finish (arrival ticket, sc1 hand-off) adds the partials in a fixed order and applies the bag
//     classifier, softmax and argmax: one launch, deterministic bits.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "pipe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int S1 = 128, S2 = 64;
constexpr float LOG2E = 1.4426950408889634f;
// floats per workgroup partial.  Without the in-kernel combine: (shift = 0, sum p, acc[128]), the layout hipt_clam_combine_launch reads;
// with it: (shift, sum p, -, -, acc[128]) -- 16-byte aligned sums, whole 16-byte loads in the merge
constexpr int PSTRIDE = 2 + S1, PSTRIDE_F = 4 + S1;

__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// sum over the 64 lanes without the LDS crossbar (ds_bpermute costs ~100 cycles a step, six dependent steps): quads and 16-lane rows by
// DPP, the four row sums by readlane; every lane gets the result; a fixed order, like wave_sum's
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    DPP_ADD(0x124);  // row_ror:4
    DPP_ADD(0x128);  // row_ror:8
#undef DPP_ADD
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
// column of accumulator register i in lane half hh of a 32 x 32 tile
__host__ __device__ __forceinline__ int acc_col(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }
// row of the stacked [Wa; Wb] matrix that gate tile t holds in its A-operand row c: tiles a[0:32] b[0:32] a[32:64] b[32:64]
__host__ __device__ __forceinline__ int gate_row(int t, int c) { return (t & 1) * S2 + (t >> 1) * 32 + c; }

// The weight image = the kernel's LDS content, in fragments of 1 KiB (one A operand of v_mfma_f32_32x32x16_bf16 per lane, 16 B):
//   W1 part:   4 bias fragments (hidden tile T), then fragment (k-step k, hidden tile T) at 4 (k + 1) + T
//   gate part: 4 bias fragments (gate tile t),   then fragment (k-step kk, gate tile t) at 4 (kk + 1) + t
//   wc [2 tile pairs][2 lane halves][16] fp32 (256 B), padding to 4 KiB
__host__ __device__ constexpr int off_wab(int KS) { return 4 * (KS + 1) * 1024; }
__host__ __device__ constexpr int off_cst(int KS) { return off_wab(KS) + 36 * 1024; }
__host__ __device__ constexpr int image_bytes(int KS) { return off_cst(KS) + 4096; }
constexpr int TB_BYTES = 32 * 144;  // a wave's transposition buffer, behind the image

// LDS-DMA as inline asm: 16 bytes per lane to (wave-uniform LDS address in M0) + 16 lane.  hipcc must not know of these loads: with its
// own builtin in flight it opens every step of the main loop with s_waitcnt vmcnt(0) (all 24 chunks of a block) instead of the counted
// per-chunk waits.  They are OLDER than every load the compiler counts (loads retire in order), so its counts stay right.
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_wave_base) {
    // (M0 is reserved: hipcc neither allocates it nor, in this kernel, uses it for anything else)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_wave_base) : "memory");
}

// fp32 -> three bf16 pieces, hi + mid + lo == v to fp32's last bit (8 + 8 + 8 mantissa bits)
__device__ __forceinline__ void split3(float v, bf16_t (&o)[3]) {
#pragma clang fp contract(off)
    o[0] = (bf16_t)v;
    const float r1 = v - (float)o[0];
    o[1] = (bf16_t)r1;
    o[2] = (bf16_t)(r1 - (float)o[1]);
}

// one thread per 16 bytes of the image
// (nb attention branches -- CLAM_MB, wc = [nb][S2] -- fill nb x 256 B of the constant area, branch k at + 256 k)
__global__ void abmil32_pack_kernel(const bf16_t* __restrict__ w1, const float* __restrict__ b1, const bf16_t* __restrict__ wab,
                                    const float* __restrict__ bab, const float* __restrict__ wc, int KS, char* __restrict__ out, int nb) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int S0 = 16 * KS, nw1 = off_wab(KS) / 16, ng = 36 * 64, total = image_bytes(KS) / 16;
    if (c >= total) return;
    u32x4 v = {0u, 0u, 0u, 0u};
    const bool gate = c >= nw1;
    const int c2 = gate ? c - nw1 : c, frag = c2 >> 6, lane = c2 & 63, r = lane & 31, hh = lane >> 5, T = frag & 3, k = (frag >> 2) - 1;
    if (c < nw1 + ng && k < 0) {
        // bias fragment: lane (r, 0) carries the bias of the tile's row r in k-slots 0..2 (the other operand is 1, 1, 1, 0, ..)
        if (hh == 0) {
            bf16_t pc[3];
            split3(gate ? bab[gate_row(T, r)] : b1[32 * T + r], pc);
            bf16x8 o = {pc[0], pc[1], pc[2], (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            v = __builtin_bit_cast(u32x4, o);
        }
    } else if (!gate) {
        // W1 fragment (k, T): lane (r, hh) <- W1[32 T + r][16 k + 8 hh ..+7]
        v = *(const u32x4*)(w1 + (int64_t)(32 * T + r) * S0 + 16 * k + 8 * hh);
    } else if (c < nw1 + ng) {
        // [Wa;Wb] fragment (kk, t): k-step kk = 2 T' + s' of hidden tile T'; lane (r, hh) element j <- hidden 16 kk + 8 (j >> 2) + 4 hh + (j & 3),
        // the order in which pack8<s'> of the h1 accumulator tile T' feeds its k slots
        const bf16_t* src = wab + (int64_t)gate_row(T, r) * S1 + 16 * k + 4 * hh;
        const u32x2 lo = *(const u32x2*)src, hi = *(const u32x2*)(src + 8);
        v = u32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
        const int f0 = (c - nw1 - ng) * 4;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f = f0 + e, tp = (f >> 5) & 1, hb = (f >> 4) & 1, i = f & 15, br = f >> 6;
            if (br < nb) o[e] = wc[br * S2 + 32 * tp + acc_col(i, hb)];
        }
        v = __builtin_bit_cast(u32x4, o);
    }
    *(u32x4*)(out + (int64_t)c * 16) = v;
}

template <int S>
__device__ __forceinline__ u32x4 pack8(const f32x16& a) {
    u32x4 o;
    o[0] = pack_bf16x2(a[8 * S + 0], a[8 * S + 1]);
    o[1] = pack_bf16x2(a[8 * S + 2], a[8 * S + 3]);
    o[2] = pack_bf16x2(a[8 * S + 4], a[8 * S + 5]);
    o[3] = pack_bf16x2(a[8 * S + 6], a[8 * S + 7]);
    return o;
}

// ---- the vector work that rides in the gaps of the MFMA streams, as lists of micro-operations dealt to the MFMA slots by issue cost
// (MI355X_MICROARCH.md: 4 cycles a vector instruction, 8 a transcendental; 24 of an MFMA's 32 cycles are free) ----
// Phase 1 carries the block before: 32 gate pairs x 12 micro-ops (tanh(x) sigmoid(y) w with ONE reciprocal:
//   acc - (w - E w) / (E (1 + F) + (1 + F)), E = e^{2x}, F = e^{-y}, x clamped to +-15), then the logit, then 16 quarter tiles of pooling.
// NB > 1 (CLAM_MB, round 6): the gate value u = tanh(x) sigmoid(y) = (E - 1) / (E (1 + F) + (1 + F)) is formed ONCE per pair and feeds one
//   multiply-add per branch (12 + NB micro-ops a pair); the logits of the NB branches follow; NO pooling here (the pooled sums of NB branches do
//   not fit the registers: a second kernel pools from the bf16 h1 this one leaves in HBM).
template <int NB> struct P1 {
    static constexpr int GU = NB == 1 ? 12 : 12 + NB, FIN = 32 * GU, POOL = FIN + 1, NU = NB == 1 ? POOL + 16 : POOL;
    static constexpr int cost(int u) {
        if (u < FIN) {
            const int j = (u % (2 * GU)) / 2;  // (two pairs interleaved, below)
            return (j == 5 || j == 6 || j == 10) ? 8 : 4;
        }
        return u == FIN ? 56 + 44 * (NB - 1) : 16;
    }
    static constexpr int total() {
        int t = 0;
        for (int u = 0; u < NU; ++u) t += cost(u);
        return t;
    }
};
// Phase 2 carries its own block's ReLU + bf16 packing: per hidden tile 8 elements, the operand of one k-step, 8 elements, the other operand
constexpr int RU = 18, P2_NU = 4 * RU;
constexpr int p2_cost(int u) { return (u % RU == 8 || u % RU == 17) ? 16 : 8; }
// micro-ops [lo, hi) of slot g when every slot takes `budget` cycles' worth: hi(g) = the first u whose running cost exceeds (g + 1) budget
template <class F>
constexpr int dealt(F cost, int nu, int budget, int g) {
    int acc = 0, u = 0;
    while (u < nu && acc + cost(u) <= (g + 1) * budget) acc += cost(u++);
    return u;
}

// Phase 1's LDS queue, in order: per slot the fragment request for slot g + PFD, then (first half of a slice) one operation on the
// transposition buffer.  lgkmcnt to wait for at slot g = the operations issued after the request of ITS fragment (4-bit counter: a
// smaller number only waits longer).
template <int KS>
constexpr int p1_lgkm(int g, int PFD) {
    constexpr int NSL = 4 * (KS + 1);
    auto nA = [&](int s) { return s + PFD < NSL ? 1 : 0; };
    auto nT = [&](int s) {
        if (s < 4) return 0;
        const int u = (s - 4) & 15;
        return (u <= 5 || u == 9 || u == 13) ? 1 : 0;
    };
    int n = 0;
    if (g >= PFD) {
        n = nT(g - PFD);
        for (int s = g - PFD + 1; s <= g; ++s) n += nA(s) + nT(s);
    } else {
        n = PFD - 1 - g;
        for (int s = 0; s <= g; ++s) n += nA(s) + nT(s);
    }
    return n < 15 ? n : 15;
}

struct Abmil32Params {
    const bf16_t* bag;
    int N, nblocks, nwaves;
    const char* image;  // the LDS image of the weights (abmil32_pack_kernel)
    const float* bc;
    float* A_raw;
    float* partials;
    int attention_only;
    unsigned* ticket;
    const float* wcls;
    const float* bcls;
    int C;
    float* M;
    float* logits;
    float* Y_prob;
    int64_t* Y_hat;
    // NB > 1 (CLAM_MB): bc = [NB], A_raw = [NB][N]; h1 = ReLU(W1 x + b1) leaves as a bf16 image for the pooling kernel below:
    // 32-row block b, piece kk (0..7) = 1 KiB at (8 b + kk) * 1024, lane (r, hh) 16 bytes: element j = hidden 16 kk + 8 (j >> 2) + 4 hh + (j & 3) of row 32 b + r
    void* h1_img;
    unsigned long long* stamps;  // diagnostic builds: [grid][24] s_memrealtime ticks (100 MHz) of wave 0
    int no_traffic;              // diagnostic builds (HIPT_ABMIL_NO_TRAFFIC): every bag request out of range -- the arithmetic alone
};

template <int KS, int NB = 1>  // k-steps of 16 input features: S0 = 16 KS; NB attention branches (1: CLAM_SB, the whole forward; > 1: CLAM_MB's first pass)
__global__ __launch_bounds__(256, 1) void abmil32_kernel(const Abmil32Params p) {
    constexpr int S0 = 16 * KS;
    constexpr int GU = P1<NB>::GU, P1_FIN = P1<NB>::FIN, P1_POOL = P1<NB>::POOL, P1_NU = P1<NB>::NU;
    auto p1_cost = [](int u) constexpr { return P1<NB>::cost(u); };
    constexpr int OFF_WAB = off_wab(KS), OFF_CST = off_cst(KS), IMG_BYTES = image_bytes(KS);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // W1 fragments | [Wa;Wb] fragments | b1, gate bias, wc in accumulator order

    const int tid = threadIdx.x, lane = tid & 63;
#define ASTAMP(k)                                                                                                                       \
    do {                                                                                                                                \
        if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 24 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
    ASTAMP(0);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int gw = blockIdx.x * 4 + w;                                  // this wave; its blocks: gw, gw + nwaves, ..
    const int nstep = gw < p.nblocks ? (p.nblocks - gw + p.nwaves - 1) / p.nwaves : 0;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.bag, 0, (int)((int64_t)p.N * S0 * 2), 0x00020000);
    constexpr int OOR = 0x7fff0000;                                     // (beyond any bag the launcher accepts)
    const int vrow = (lane >> 3) * (S0 * 2) + (lane & 7) * 16;          // lane l: row l >> 3 of a group of 8, chunk l & 7 of a 128-byte line
    const __amdgpu_buffer_rsrc_t h1rs = __builtin_amdgcn_make_buffer_rsrc(p.h1_img, 0, NB > 1 ? p.nblocks * 8192 : 0, 0x00020000);

    // ---- stage the weights: the image is the LDS content byte for byte, 1 KiB per LDS-DMA wave instruction ----
    {
        constexpr int PER_WAVE = IMG_BYTES / 4096;  // KiB per wave
        const char* src = p.image + (w * PER_WAVE) * 1024 + lane * 16;
        char* dst = smem + (w * PER_WAVE) * 1024;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) glds16_asm(src + j * 1024, lds_addr(dst) + j * 1024);
    }
    // this wave's first block: requested behind the image, so that "all but the youngest KS" below means "the image has landed"
    // (the bag through a buffer resource that ends with it: a chunk of a row past the end reads as zero, no traffic)
    // The block in flight, in line layout: xc[4 i + j] = lane l's 16 bytes of row 8 j + (l >> 3), 128-byte line i, chunk l & 7
    constexpr int NSLICE = KS / 4;
    u32x4 xc[KS];
    auto xoff = [](int m) { return (m & 3) * 8 * (S0 * 2) + (m >> 2) * 128; };
    {
        const int v0 = nstep > 0 && !p.no_traffic ? gw * 32 * S0 * 2 + vrow : OOR;  // (no block: out of range, zeros)
#pragma unroll
        for (int m = 0; m < KS; ++m) xc[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, v0 + xoff(m), 0, 2);
    }
    // s_waitcnt vmcnt(KS) as the builtin (vmcnt = bits 15:14 | 3:0; expcnt, lgkmcnt: no wait): hipcc's own wait-count bookkeeping sees
    // it -- behind an opaque asm wait it would take every load as still in flight and open each step with vmcnt(0)
    __builtin_amdgcn_s_waitcnt(((KS >> 4) << 14) | 0x0F70 | (KS & 15));
    ASTAMP(1);
    __builtin_amdgcn_s_barrier();  // (raw: __syncthreads() would wait for the first block's loads too)  weights are in LDS; from here on the waves never synchronise again (until the merge)

    const uint32_t lbase = lds_addr(smem);
    // (ds offsets are 16-bit: three bases for the 100 + 36 fragments)
    const uint32_t fa = lbase + lane * 16, fb = fa + 64 * 1024, fg = fa + OFF_WAB;
    // wc of this lane's 32 gate pairs, in pair order (tile pair tp, register i): kept in registers
    // (through asm reads: a visible LDS access makes hipcc wait for every load in flight, the first block's included)
    float wcr[NB][32];
    sfor<0, NB>([&](auto K_) __attribute__((always_inline)) {
        constexpr int k = decltype(K_)::value;
        const uint32_t a = lbase + OFF_CST + hh * 64;  // [branch][tile pair][lane half][16]: branch k at + 256 k, pair tp at + 128 tp
        f32x4 v[8];
        DSR128X4_WAIT(v[0], v[1], v[2], v[3], a, 256 * k + 0, 256 * k + 16, 256 * k + 32, 256 * k + 48);
        DSR128X4_WAIT(v[4], v[5], v[6], v[7], a, 256 * k + 128, 256 * k + 144, 256 * k + 160, 256 * k + 176);
#pragma unroll
        for (int q4 = 0; q4 < 8; ++q4) {
            wcr[k][4 * q4] = v[q4][0];
            wcr[k][4 * q4 + 1] = v[q4][1];
            wcr[k][4 * q4 + 2] = v[q4][2];
            wcr[k][4 * q4 + 3] = v[q4][3];
        }
    });
    float bcv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) bcv[k] = p.bc[k];
    // the other operand of the bias k-steps: 1, 1, 1, 0, .. in the k-slots of lane half 0
    const u32x4 ones = hh == 0 ? u32x4{0x3f803f80u, 0x00003f80u, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    f32x16 pool[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) pool[T] = zero16;
    float lsum = 0.f;
    f32x16 H[4], G[4];   // accumulators of the two products
    f32x16 Hp[4];        // h1 = ReLU(H) in fp32, kept for the pooling one block later
    u32x4 hf[8];         // h1 in bf16: the B operand of the gate product
    float gs = 0.f, prow = 0.f;
    float gsk[NB];  // NB > 1: the logits of the branches, in the making
#pragma unroll
    for (int k = 0; k < NB; ++k) gsk[k] = 0.f;

    // ---- micro-ops (above) ----
    // two gate pairs in flight, their micro-ops alternating: one wave per SIMD, so a dependent chain of vector instructions has nobody
    // to hide its latencies behind (a transcendental's result is not ready for the next instruction) but the other pair
    float ga[2], gb[2], xs[2], ys[2], eE[2], eF[2], nn[2], t1[2], dn[2], rc[2];
    auto finish = [&](int blk) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        if constexpr (NB > 1) {
            const int row = blk * 32 + r;
            const bool valid = row < p.N && blk >= 0;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const unsigned gbits = __builtin_bit_cast(unsigned, gsk[k]);
                const auto sw = __builtin_amdgcn_permlane32_swap(gbits, gbits, false, false);
                const unsigned s0 = sw[0], s1 = sw[1];
                const float g2 = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
                if (valid && hh == 0) p.A_raw[(int64_t)k * p.N + row] = g2 + bcv[k];
            }
            return;
        }
        // the row's logit (the lane halves hold the two halves of its gate units), A_raw, softmax weight against the fixed shift
        // v_permlane32_swap on two copies: one becomes (lower, lower), the other (upper, upper): the same sum, in the same order, in
        // both halves (the pooling of either half uses prow); no LDS crossbar in the MFMA stream
        const unsigned gbits = __builtin_bit_cast(unsigned, gs);
        const auto sw = __builtin_amdgcn_permlane32_swap(gbits, gbits, false, false);
        const unsigned s0 = sw[0], s1 = sw[1];  // (scalar copies first: bit-casting a vector element reads element 0, common.h)
        const float g2 = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
        const int row = blk * 32 + r;
        const bool valid = row < p.N && blk >= 0;
        if (valid && hh == 0) p.A_raw[row] = g2 + bcv[0];
        prow = valid ? __builtin_amdgcn_exp2f(g2 * LOG2E) : 0.f;  // e^(A - bc), in [e^-B, e^B]
        if (hh == 0) lsum += prow;
    };
    auto p1_uop = [&](auto U_, int blk) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        constexpr int u = decltype(U_)::value;
        if constexpr (u < P1_FIN) {
            // gate pair q = 16 tp + i: a = G[2 tp][i], b = G[2 tp + 1][i] (biases in).  Every multiply-add is written out (contraction off: a
            // row's logit must be the same bits wherever the row sits -- pipelined step, drain, any block).  The accumulators live in the
            // accumulator file and are read HERE: left to hipcc, all 64 reads of a block sit in one burst in front of the next block's MFMAs
            constexpr int c = u & 1, q = 2 * (u / (2 * GU)) + c, j = (u % (2 * GU)) / 2, tp = q >> 4, i = q & 15;
            if constexpr (j == 0) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ga[c]) : "a"(G[2 * tp][i]));
            if constexpr (j == 1) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(gb[c]) : "a"(G[2 * tp + 1][i]));
            if constexpr (j == 2) xs[c] = __builtin_amdgcn_fmed3f(ga[c], -15.0f, 15.0f);
            if constexpr (j == 3) xs[c] = xs[c] * (2.0f * LOG2E);
            if constexpr (j == 4) ys[c] = gb[c] * -LOG2E;
            if constexpr (j == 5) eE[c] = __builtin_amdgcn_exp2f(xs[c]);
            if constexpr (j == 6) eF[c] = __builtin_amdgcn_exp2f(ys[c]);
            // (w - E w = -numerator: the negations sit on E and on the product as source modifiers; a "-w" would be a loop invariant that
            //  hipcc hoists into 32 more live registers)
            if constexpr (NB == 1) {
                if constexpr (j == 7) nn[c] = __builtin_fmaf(-eE[c], wcr[0][q], wcr[0][q]);
                if constexpr (j == 11) gs = __builtin_fmaf(-nn[c], rc[c], q == 0 ? 0.f : gs);  // (pair order: 0, 1, 2, ..)
            } else {
                // several branches: u = (E - 1) / (E (1 + F) + (1 + F)) once, then one multiply-add per branch (nn = E - 1, then nn = u)
                if constexpr (j == 7) nn[c] = eE[c] - 1.0f;
                if constexpr (j == 11) nn[c] = nn[c] * rc[c];
                if constexpr (j >= 12) gsk[j - 12] = __builtin_fmaf(nn[c], wcr[j - 12 < NB ? j - 12 : 0][q], q == 0 ? 0.f : gsk[j - 12]);
            }
            if constexpr (j == 8) t1[c] = eF[c] + 1.0f;
            if constexpr (j == 9) dn[c] = __builtin_fmaf(eE[c], t1[c], t1[c]);
            if constexpr (j == 10) rc[c] = __builtin_amdgcn_rcpf(dn[c]);
        } else if constexpr (u == P1_FIN) {
            finish(blk);
        } else if constexpr (NB == 1) {
            constexpr int T = (u - P1_POOL) >> 2, q4 = (u - P1_POOL) & 3;  // a quarter of a tile: 4 registers
#pragma unroll
            for (int i = 4 * q4; i < 4 * q4 + 4; ++i) pool[T][i] = __builtin_fmaf(prow, Hp[T][i], pool[T][i]);
        }
    };
    auto p2_uop = [&](auto U_) __attribute__((always_inline)) {
        constexpr int u = decltype(U_)::value, T = u / RU, j = u % RU;
        if constexpr (j == 8) hf[2 * T] = pack8<0>(Hp[T]);
        else if constexpr (j == 17) hf[2 * T + 1] = pack8<1>(Hp[T]);
        else {
            constexpr int e = j < 8 ? j : j - 1;
            Hp[T][e] = fmaxf(H[T][e], 0.f);
        }
    };

    // the wave's transposition buffer: 32 rows of 128 + 16 bytes (the padding spreads the rows over the banks for the operand reads)
    const uint32_t tb = lbase + IMG_BYTES + w * TB_BYTES;
    const uint32_t tbw = tb + (lane >> 3) * 144 + (lane & 7) * 16;  // line layout: + 8 j rows
    const uint32_t tbr = tb + r * 144 + hh * 16;                     // operand layout: lane (r, hh), k-step kk of the slice: + 32 kk
    u32x4 xr[4];     // the B operands of a slice's 4 k-steps; each is replaced by the next slice's as soon as its k-step is done
    // slice i of the block in xc -> the buffer, and its registers re-requested for the wave's next block (the store has read them)
    auto tb_write = [&](auto M_, int vnext) __attribute__((always_inline)) {
        constexpr int m = decltype(M_)::value, j = m & 3;
        const uint32_t a = tbw;
        const u32x4 d = xc[m];  // (plain copies first: a generic lambda does not capture what is only an asm operand)
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "a"(d), "n"(j * 8 * 144) : "memory");
        xc[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, vnext + xoff(m), 0, 2);
    };
    auto tb_read = [&](auto KK_) __attribute__((always_inline)) {
        constexpr int kk = decltype(KK_)::value;
        const uint32_t a = tbr;
        u32x4 d;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(a), "n"(kk * 32));
        xr[kk] = d;
    };

    // fragment read into the ring (slot g of a phase: fragment index frag0 + g)
    constexpr int NWF = 6;  // ring entries: PFD in flight + the one in use
    u32x4 wfr[NWF];
    auto rd1 = [&](auto G_) __attribute__((always_inline)) {  // phase 1: fragment g
        constexpr int g = decltype(G_)::value;
        u32x4& d = wfr[g % NWF];
        if constexpr (g < 64) {
            const uint32_t a = fa;
            DSR128(d, a, g * 1024);
        } else {
            const uint32_t a = fb;
            DSR128(d, a, (g - 64) * 1024);
        }
    };
    auto rd2 = [&](auto G_) __attribute__((always_inline)) {  // phase 2: gate fragment g
        constexpr int g = decltype(G_)::value;
        u32x4& d = wfr[g % NWF];
        const uint32_t a = fg;
        DSR128(d, a, g * 1024);
    };
    constexpr int PFD = 5;  // fragments requested ahead of the MFMA that uses them

    // phase 1 of a block: H = W1 x^T + b1: the bias k-step, then NSLICE slices of 4 k-steps, 4 hidden tiles each.  The transposition
    // buffer holds the slice being multiplied until its last operand has been read (position 0 of the slice), then takes the next slice
    // (positions 1 .. 4: stores + re-requests), whose operands replace this slice's one by one as their k-steps finish (positions 5, 9,
    // 13 and position 0 of the next slice) -- in order in the wave's LDS queue, so no waits.  Under the last slice it is slice 0 of the
    // wave's NEXT block (its lines were re-requested a block ago).  With PIPE the gate arithmetic, the logit and the pooling of the
    // block BEFORE (in G and Hp) ride in the gaps of the MFMA stream.
    auto phase1 = [&](auto PIPE_, int vnext, int vnext2, int blk_prev) __attribute__((always_inline)) {
        constexpr bool PIPE = decltype(PIPE_)::value;
        constexpr int NSL = 4 * (KS + 1);
        constexpr int BUD = (P1<NB>::total() + NSL - 1) / NSL > 24 ? (P1<NB>::total() + NSL - 1) / NSL : 24;  // cycles of vector work per slot
        sfor<0, PFD>(rd1);
        sfor<0, NSL>([&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value, k = (g >> 2) - 1, T = g & 3;
            constexpr int sl = k < 0 ? -1 : k >> 2, u = g < 4 ? -1 : (g - 4) & 15;   // slice of this slot, position in it
            if constexpr (g + PFD < NSL) rd1(std::integral_constant<int, g + PFD>{});
            if constexpr (u == 0) tb_read(std::integral_constant<int, 3>{});
            // (slices 1 .. of this block free their registers for the next block; slice 0 of the NEXT block frees them for the one after)
            if constexpr (u >= 1 && u <= 4) tb_write(std::integral_constant<int, 4 * ((sl + 1) % NSLICE) + u - 1>{}, sl + 1 < NSLICE ? vnext : vnext2);
            if constexpr (u == 5 || u == 9 || u == 13) tb_read(std::integral_constant<int, (u - 5) / 4>{});
            LGKM(p1_lgkm<KS>(g, PFD));
            if constexpr (k < 0) H[T] = mfma32(wfr[g % NWF], ones, zero16);
            else H[T] = mfma32(wfr[g % NWF], xr[k & 3], H[T]);
            if constexpr (PIPE) {
                constexpr int lo = g == 0 ? 0 : dealt(p1_cost, P1_NU, BUD, g - 1), hi = g == NSL - 1 ? P1_NU : dealt(p1_cost, P1_NU, BUD, g);
                sfor<lo, hi>([&](auto U_) __attribute__((always_inline)) { p1_uop(U_, blk_prev); });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // phase 2: G = [Wa;Wb] h1^T + gate bias: the bias k-step, then 8 k-steps, 4 gate tiles each.  The ReLU / packing of h1 rides along: tile
    // 0 under the bias k-step, tile T + 1 under the two k-steps that consume tile T.
    auto phase2 = [&]() __attribute__((always_inline)) {
        constexpr int NSL = 36;
        sfor<0, PFD>(rd2);
        sfor<0, NSL>([&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value, kk = (g >> 2) - 1, t = g & 3;
            if constexpr (g + PFD < NSL) {
                rd2(std::integral_constant<int, g + PFD>{});
                LGKM(PFD);
            } else {
                LGKM(NSL - 1 - g);
            }
            if constexpr (kk < 0) G[t] = mfma32(wfr[g % NWF], ones, zero16);
            else G[t] = mfma32(wfr[g % NWF], hf[kk], G[t]);
            // (24 cycles' worth a slot: the operand of k-step kk = 2 T + s is packed 80 (2 T + s + 1) cycles into the list, k-step kk starts at
            //  slot 4 (kk + 1): always in time)
            constexpr int lo = g == 0 ? 0 : dealt(p2_cost, P2_NU, 24, g - 1), hi = dealt(p2_cost, P2_NU, 24, g);
            sfor<lo, hi>(p2_uop);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // gate arithmetic + pooling of the wave's last block: nothing left to hide them under
    auto drain = [&](int blk) __attribute__((always_inline)) {
        sfor<0, P1_NU>([&](auto U_) __attribute__((always_inline)) { p1_uop(U_, blk); });
    };

    auto step = [&](auto PIPE_, int s) __attribute__((always_inline)) {
        const int blk = gw + s * p.nwaves;
        const int vnext = s + 1 < nstep && !p.no_traffic ? (blk + p.nwaves) * 32 * S0 * 2 + vrow : OOR;
        const int vnext2 = s + 2 < nstep && !p.no_traffic ? (blk + 2 * p.nwaves) * 32 * S0 * 2 + vrow : OOR;
        phase1(PIPE_, vnext, vnext2, blk - p.nwaves);
        if (s < 4) ASTAMP(3 + 3 * s);
        phase2();
        if constexpr (NB > 1) {
            // this block's h1 in bf16 (the gate product's operand registers: complete behind phase 2) -> the image the pooling kernel reads:
            // eight stores of 1 KiB per wave, consecutive bytes
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) __builtin_amdgcn_raw_buffer_store_b128(hf[kk], h1rs, (blk * 8 + kk) * 1024 + lane * 16, 0, 0);
        }
        if (s < 4) ASTAMP(4 + 3 * s);
    };
    ASTAMP(2);
    if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 24 + 21] = __builtin_amdgcn_s_memtime();
    if (nstep > 0) {
        {   // the first block's slice 0 (later blocks: under the last slice of the block before)
            const int v1 = nstep > 1 && !p.no_traffic ? (gw + p.nwaves) * 32 * S0 * 2 + vrow : OOR;
            sfor<0, 4>([&](auto J_) __attribute__((always_inline)) { tb_write(J_, v1); });
            sfor<0, 3>(tb_read);
        }
        // one loop body for every block: the first carries the (masked) vector work of a block that does not exist -- a separate plain
        // first step costs hipcc ~70 registers spilled and reloaded around the loop
#pragma unroll
        for (int T = 0; T < 4; ++T) G[T] = Hp[T] = zero16;
        for (int s = 0; s < nstep; ++s) step(std::true_type{}, s);
        drain(gw + (nstep - 1) * p.nwaves);
    }
    ASTAMP(15);
    if (HIPT_STAMPS_ON(p.stamps) && threadIdx.x == 0) p.stamps[(size_t)blockIdx.x * 24 + 22] = __builtin_amdgcn_s_memtime();
    if (p.attention_only || NB > 1) return;  // (several branches: the pooling kernel below takes over)

    // ---- this workgroup's partial: sum over the rows (= lanes) of every wave, then over the 4 waves, through LDS (weights are dead) ----
    __syncthreads();
    constexpr int RS = S1 + 4;   // row stride in floats: 16 bytes of padding, or the 32 lanes of a store all hit the same banks
    float* red = (float*)smem;  // [4 waves][32 rows][RS] fp32 = 66 KiB | lsum [4][32] behind it
    {
        float* dst = red + ((w * 32 + r) * RS);
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int q = 0; q < 4; ++q)  // registers 4 q ..+3 = hidden 32 T + 8 q + 4 hh ..+3
                *(f32x4*)(dst + 32 * T + 8 * q + 4 * hh) = f32x4{pool[T][4 * q], pool[T][4 * q + 1], pool[T][4 * q + 2], pool[T][4 * q + 3]};
        if (hh == 0) red[4 * 32 * RS + w * 32 + r] = lsum;
    }
    __syncthreads();
    const bool fused = p.ticket != nullptr;
    float* pw = p.partials + (int64_t)blockIdx.x * (fused ? PSTRIDE_F : PSTRIDE);
    float* pacc = pw + (fused ? 4 : 2);
    {
        // thread (half, col): rows [64 half, 64 half + 64) of column col, in row order; the two halves are added by half 0
        const int col = tid & 127, half = tid >> 7;
        // (four chains of 16, added in a fixed order: one chain of 64 is 64 dependent additions behind their LDS reads)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a0 += red[(half * 64 + j) * RS + col];
            a1 += red[(half * 64 + 16 + j) * RS + col];
            a2 += red[(half * 64 + 32 + j) * RS + col];
            a3 += red[(half * 64 + 48 + j) * RS + col];
        }
        const float a = (a0 + a1) + (a2 + a3);
        float l2 = 0.f;
        if (col < 64) l2 = red[4 * 32 * RS + half * 64 + col];
        l2 = wave_sum_dpp(l2);  // (waves 0, 1: half 0; waves 2, 3: half 1; only the waves with col < 64 hold values)
        __syncthreads();
        float* ex = red;  // exchange: [2][128] column sums, [4] wave sums of l
        ex[half * S1 + col] = a;
        if (lane == 0) ex[256 + w] = l2;
        __syncthreads();
        // (agent-scope relaxed stores = sc1 stores: they leave the XCD's L2, the merging workgroup reads them with sc1 loads and no fence)
        if (tid < S1) __hip_atomic_store(&pacc[tid], ex[tid] + ex[S1 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(&pw[0], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every partial against the same shift
            __hip_atomic_store(&pw[1], (ex[256] + ex[257]) + (ex[258] + ex[259]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- fused combine (model_clam.py:180-183): the workgroup whose ticket is the last one adds all partials in a fixed order,
    //      applies the bag classifier, softmax and argmax.  Hand-off without fences (MI355X_MICROARCH.md, hand-off table row 1): sc1 stores,
    //      every storing wave waits vmcnt(0), workgroup barrier, ONE agent-scope atomic per workgroup; the workgroup whose add came
    //      last reads with sc1 loads after a workgroup barrier.  The ticket starts at zero and the last arriver puts it back.
    if (!fused) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ASTAMP(16);
    __syncthreads();
    int* flag = (int*)(red + 512);
    if (tid == 0) {
        const bool last = atomicAdd(p.ticket, 1u) == gridDim.x - 1;
        *flag = last;
        if (last) __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    ASTAMP(17);
    if (!*flag) return;
    {
        const int Gn = gridDim.x;
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)p.partials, 0, Gn * PSTRIDE_F * 4, 0x00020000);
        constexpr int SC1 = 16;
        // 32 threads x 16 B cover the 128 sums of one partial, 8 partials per pass; all of a thread's share is requested at once
        // (rows past Gn are out of the buffer's range and read as zero); summed in partial order: deterministic
        const int c4 = tid & 31, part = tid >> 5;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        f32x4 rowv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            rowv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, ((part + 8 * i) * PSTRIDE_F + 4 + 4 * c4) * 4, 0, SC1));
        float lv = tid < Gn ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, tid * PSTRIDE_F * 4 + 4, 0, SC1)) : 0.f;
        // the classifier (<= 8 classes: wave 0 keeps all of it in registers, requested now, with the partials -- one round trip, not two)
        constexpr int CF = 8;
        const bool small_c = p.C <= CF;
        float wca[CF], wcb[CF], bcl[CF];
#pragma unroll
        for (int k = 0; k < CF; ++k) {
            const bool on = small_c && w == 0 && k < p.C;
            wca[k] = on ? p.wcls[(int64_t)k * S1 + lane] : 0.f;
            wcb[k] = on ? p.wcls[(int64_t)k * S1 + lane + 64] : 0.f;
            bcl[k] = on ? p.bcls[k] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) a += rowv[i];
        ASTAMP(18);
        float* Cs = red + 1024;   // [8][128] column partial sums
        float* Ms = red + 2048;   // [128]
        float* Ls = red + 2176;   // [C <= 64]
        float* wr = red + 2240;   // [4] wave sums of l
        *(f32x4*)(Cs + part * S1 + 4 * c4) = a;
        lv = wave_sum_dpp(lv);
        if (lane == 0) wr[w] = lv;
        __syncthreads();
        const float L = (wr[0] + wr[1]) + (wr[2] + wr[3]);
        if (tid < S1) {
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) m += Cs[q * S1 + tid];
            m /= L;
            Ms[tid] = m;
            p.M[tid] = m;
        }
        __syncthreads();
        ASTAMP(19);
        if (small_c) {
            if (w != 0) return;
            const float m0 = Ms[lane], m1 = Ms[lane + 64];
            float lg[CF];
            float lm = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int k = 0; k < CF; ++k) {
                lg[k] = 0.f;
                if (k < p.C) {  // (uniform)
                    lg[k] = wave_sum_dpp(m0 * wca[k] + m1 * wcb[k]) + bcl[k];
                    if (lg[k] > lm) {
                        lm = lg[k];
                        arg = k;
                    }
                }
            }
            float ev[CF], se = 0.f;
#pragma unroll
            for (int k = 0; k < CF; ++k) {
                ev[k] = 0.f;
                if (k < p.C) {
                    ev[k] = __builtin_amdgcn_exp2f((lg[k] - lm) * LOG2E);  // (<= 0: no range issue; 1 ulp of v_exp_f32)
                    se += ev[k];
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < CF; ++k)
                    if (k < p.C) {
                        p.logits[k] = lg[k];
                        p.Y_prob[k] = ev[k] / se;
                    }
                p.Y_hat[0] = arg;
            }
        } else {
            for (int k = w; k < p.C; k += 4) {
                float v = Ms[lane] * p.wcls[(int64_t)k * S1 + lane] + Ms[lane + 64] * p.wcls[(int64_t)k * S1 + lane + 64];
                v = wave_sum_dpp(v);
                if (lane == 0) Ls[k] = v + p.bcls[k];
            }
            __syncthreads();
            if (tid == 0) {
                float lm = -INFINITY;
                int arg = 0;
                for (int k = 0; k < p.C; ++k)
                    if (Ls[k] > lm) {
                        lm = Ls[k];
                        arg = k;
                    }
                float se = 0.f;
                for (int k = 0; k < p.C; ++k) se += expf(Ls[k] - lm);
                for (int k = 0; k < p.C; ++k) {
                    p.logits[k] = Ls[k];
                    p.Y_prob[k] = expf(Ls[k] - lm) / se;
                }
                p.Y_hat[0] = arg;
            }
        }
        ASTAMP(20);
    }
}

template <int KS>
int launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* partials, int* n_partials,
           unsigned* ticket, float* M, float* logits, float* Y_prob, int64_t* Y_hat, hipStream_t st) {
    constexpr int lds = image_bytes(KS) + 4 * TB_BYTES;
    constexpr int lds_alloc = lds > 72 * 1024 ? lds : 72 * 1024;  // (the reduction at the end uses 66.5 KiB of it)
    auto k = abmil32_kernel<KS>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_alloc) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(abmil32) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("abmil32: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    Abmil32Params p;
    p.bag = (const bf16_t*)bag;
    p.N = N;
    p.nblocks = (N + 31) / 32;
    // every wave the same number of blocks (+-1): 100 000 rows are 3125 blocks = 4 rounds of 782 waves (196 workgroups), not 3 rounds
    // of 1024 and a fourth of 53 -- the time is the slowest wave's either way, and the even spread asks less of the HBM per round
    const int maxg = once.ncu[dev] < 256 ? once.ncu[dev] : 256;  // (the merge reads up to 256 partials in one round trip)
    const int rounds = (p.nblocks + 4 * maxg - 1) / (4 * maxg);
    const int grid = ((p.nblocks + rounds - 1) / rounds + 3) / 4;
    p.nwaves = grid * 4;
    p.image = (const char*)w->stream_pk;
    p.bc = w->bc;
    p.A_raw = A_raw;
    p.partials = partials;
    p.attention_only = attention_only;
    const bool fuse = !attention_only && ticket && M && w->n_classes <= 64;
    p.ticket = fuse ? ticket : nullptr;
    p.wcls = w->wcls;
    p.bcls = w->bcls;
    p.C = w->n_classes;
    p.M = M;
    p.logits = logits;
    p.Y_prob = Y_prob;
    p.Y_hat = Y_hat;
    p.h1_img = nullptr;  // (one branch: nothing leaves but A_raw and the pooled result)
    p.stamps = nullptr;
    p.no_traffic = 0;
#ifdef HIPT_DEBUG_STAMPS
    static const bool no_traffic = getenv("HIPT_ABMIL_NO_TRAFFIC") != nullptr;
    p.no_traffic = no_traffic;  // diagnostic builds only (make DEBUG_STAMPS=1): the release library never allocates or synchronises
    static const bool want_stamps = getenv("HIPT_ABMIL_STAMPS") != nullptr;
    static unsigned long long* dbuf = nullptr;
    if (want_stamps && !dbuf) (void)hipMalloc(&dbuf, 256 * 24 * sizeof(unsigned long long));
    if (want_stamps) {
        (void)hipMemsetAsync(dbuf, 0, 256 * 24 * sizeof(unsigned long long), st);
        p.stamps = dbuf;
    }
#endif
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_alloc, st, p);
    HIPT_CHECK_LAUNCH();
#ifdef HIPT_DEBUG_STAMPS
    if (want_stamps) {
        static unsigned long long h[256 * 24];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(h, dbuf, (size_t)grid * 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < grid; ++b)
            if (h[b * 24] < t0) t0 = h[b * 24];
        // per stamp: mean and maximum over the workgroups of (stamp - earliest start), us
        fprintf(stderr, "[abmil32 N=%d grid=%d] stamp: mean / max us since the first workgroup started (0 start, 1 weights staged, 2 loop, 3+3s phase 1 of step s, 4+3s its gate GEMM, 15 loop end incl. the drain, 16 partial stored, 17 ticket taken; last workgroup: 18 partials read, 19 M, 20 done)\n", N, grid);
        {   // shader clock over the main loop: s_memtime ticks (stamps 21, 22) per 100 MHz tick (stamps 2, 15)
            double f = 0;
            for (int b = 0; b < grid; ++b) f += (double)(h[b * 24 + 22] - h[b * 24 + 21]) / (double)(h[b * 24 + 15] - h[b * 24 + 2]) * 0.1 / grid;
            fprintf(stderr, "   shader clock in the main loop: %.2f GHz\n", f);
        }
        for (int k2 = 0; k2 < 21; ++k2) {
            double sum = 0, mx = 0;
            int n = 0;
            for (int b = 0; b < grid; ++b) {
                if (!h[b * 24 + k2]) continue;
                const double d = (double)(h[b * 24 + k2] - t0) * 0.01;
                sum += d;
                mx = d > mx ? d : mx;
                ++n;
            }
            if (n) fprintf(stderr, "   %2d: %7.2f / %7.2f  (%d workgroups)\n", k2, sum / n, mx, n);
        }
    }
#endif
    *n_partials = fuse ? 0 : grid;  // 0: the kernel has already produced M / logits / Y_prob / Y_hat
    return HIPT_OK;
}

// ---- CLAM_MB, second pass (round 6): M[k] = softmax_N(A_raw[k]) h1 for the NB branches from the bf16 h1 image and the logits the first pass left
// (models/model_clam.py:233-250).  A wave walks 32-row blocks (block b -> wave b mod #waves): 8 KiB of h1 per block in eight 1 KiB loads, the
// rows' NB softmax weights e^(A - bc) (the fixed shift of the first pass: |A - bc| <= sum |wc| < 60), 64 x NB multiply-adds per lane; then the
// workgroup's partial per branch through LDS (the reduction of abmil32_kernel), and the last workgroup to arrive (ticket) adds the partials in
// a fixed order and applies the K one-row classifiers.  Deterministic bits.
struct ClamMbPoolParams {
    const char* h1_img;
    const float* A_raw;   // [NB][N]
    const float* bc;      // [NB]
    int N, nblocks;
    float* partials;      // [grid][NB][PSTRIDE_F]
    unsigned* ticket;
    const float* wcls;    // [NB][S1]
    const float* bcls;    // [NB]
    float* M;             // [NB][S1]
    float* logits;        // [NB]
};

// sum over the 32 lanes of a lane half (every lane of the half gets it): quads and 16-lane rows by DPP, the two rows by v_permlane16_swap
__device__ __forceinline__ float half_sum_dpp(float v) {
#pragma clang fp contract(off)
#define DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    DPP_ADD(0xB1);   // quad_perm [1,0,3,2]
    DPP_ADD(0x4E);   // quad_perm [2,3,0,1]
    DPP_ADD(0x124);  // row_ror:4
    DPP_ADD(0x128);  // row_ror:8
#undef DPP_ADD
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __builtin_bit_cast(float, (uint32_t)s16[0]) + __builtin_bit_cast(float, (uint32_t)s16[1]);
}

// Eight waves per workgroup; the workgroup walks 32-row blocks (b = blockIdx.x, += gridDim.x) and wave w owns PIECE w of every block (1 KiB: the
// hidden units 16 w .. 16 w + 15 of the block's 32 rows): 8 x NB running sums per lane instead of 64 x NB, so that eight blocks' loads are in
// flight per wave (the first version walked whole blocks per wave with nothing in flight behind the block in work: 20 us for 26 MB).
template <int NB>
__global__ __launch_bounds__(512) void clam_mb_pool_kernel(const ClamMbPoolParams p) {
    __shared__ float sh[64];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int G = gridDim.x;
    float acc[NB][8];
    float ls[NB], bcv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        ls[k] = 0.f;
        bcv[k] = p.bc[k];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    }
    constexpr int PFD = 8;  // blocks in flight per wave
    const int nmine = blockIdx.x < p.nblocks ? (p.nblocks - blockIdx.x + G - 1) / G : 0;
    // (range-checked buffer loads: a block past the end reads as zero without traffic and without a branch -- hipcc drains every load in flight
    //  at a conditional one)
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.h1_img, 0, p.nblocks * 8192, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)p.A_raw, 0, (int)((int64_t)NB * p.N * 4), 0x00020000);
    constexpr int OOR = 0x7fff0000;
    const int lane_off = w * 1024 + lane * 16;
    auto ld = [&](int i) { return __builtin_amdgcn_raw_buffer_load_b128(hrs, i < nmine ? (blockIdx.x + i * G) * 8192 + lane_off : OOR, 0, 0); };
    auto lda = [&](int i, int k) {
        const int row = (blockIdx.x + i * G) * 32 + r;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ars, (i < nmine && row < p.N) ? (k * p.N + row) * 4 : OOR, 0, 0));
    };
    u32x4 win[PFD];
    float aw[PFD][NB];
#pragma unroll
    for (int d = 0; d < PFD; ++d) {
        win[d] = ld(d);
#pragma unroll
        for (int k = 0; k < NB; ++k) aw[d][k] = lda(d, k);
    }
    for (int i0 = 0; i0 < nmine; i0 += PFD) {
#pragma unroll
        for (int d = 0; d < PFD; ++d) {
#pragma clang fp contract(off)
            const u32x4 h = win[d];
            const bool valid = i0 + d < nmine && (blockIdx.x + (i0 + d) * G) * 32 + r < p.N;
            float pk[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const float e = __builtin_amdgcn_exp2f((aw[d][k] - bcv[k]) * LOG2E);
                pk[k] = valid ? e : 0.f;  // (a row that does not exist weighs nothing)
                ls[k] += pk[k];
            }
            win[d] = ld(i0 + d + PFD);
#pragma unroll
            for (int k = 0; k < NB; ++k) aw[d][k] = lda(i0 + d + PFD, k);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t wd = h[e >> 1];
                const float v = __builtin_bit_cast(float, (e & 1) ? (wd & 0xffff0000u) : (wd << 16));
#pragma unroll
                for (int k = 0; k < NB; ++k) acc[k][e] = __builtin_fmaf(pk[k], v, acc[k][e]);
            }
        }
    }
    // ---- this workgroup's partial: per value the sum over the 32 rows of its lane half; element e of lane half hh = hidden 16 w + 8 (e >> 2) + 4 hh + (e & 3) ----
    float* pw = p.partials + (int64_t)blockIdx.x * NB * PSTRIDE_F;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sme = half_sum_dpp(acc[k][e]);
            if (r == 0) __hip_atomic_store(&pw[k * PSTRIDE_F + 4 + 16 * w + 8 * (e >> 2) + 4 * hh + (e & 3)], sme, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float l = half_sum_dpp(ls[k]);  // (every wave walked the same rows: wave 0, lane half 0 speaks for all)
        if (w == 0 && lane == 0) __hip_atomic_store(&pw[k * PSTRIDE_F + 1], l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- hand-off as in abmil32_kernel: sc1 stores, every wave's vmcnt(0), barrier, one agent-scope atomic; the last arriver merges ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const bool last = atomicAdd(p.ticket, 1u) == gridDim.x - 1;
        flag = last;
        if (last) __hip_atomic_store(p.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!flag) return;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)p.partials, 0, G * NB * PSTRIDE_F * 4, 0x00020000);
    constexpr int SC1 = 16;
    // thread (part, col): column col of the workgroups g = part, part + 4, .. (G <= 128: 32 of them), ALL its loads in flight at once (the first
    // version took them eight at a time: sixteen round trips); summed in ascending g, then the four parts in order: deterministic bits
    __shared__ float cs[NB][4][S1];
    __shared__ float lsh[NB][8];
    const int col = tid & 127, part = tid >> 7;
    float v[NB][32];
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int g = part + 4 * i;
            v[k][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, g < G ? ((g * NB + k) * PSTRIDE_F + 4 + col) * 4 : OOR, 0, SC1));
        }
    float lv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) lv[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, tid < G ? ((tid * NB + k) * PSTRIDE_F + 1) * 4 : OOR, 0, SC1));
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) a += v[k][i];
        cs[k][part][col] = a;
        const float l = wave_sum_dpp(lv[k]);
        if (lane == 0) lsh[k][w] = l;
    }
    __syncthreads();
    const int k = tid >> 7;  // thread (k, col) finishes column col of branch k
    float m = 0.f;
    if (k < NB) {
        const float L = ((lsh[k][0] + lsh[k][1]) + (lsh[k][2] + lsh[k][3])) + ((lsh[k][4] + lsh[k][5]) + (lsh[k][6] + lsh[k][7]));
        m = ((cs[k][0][col] + cs[k][1][col]) + (cs[k][2][col] + cs[k][3][col])) / L;
        p.M[k * S1 + col] = m;
    }
    // logits[k] = wcls[k] . M[k] + bcls[k] (model_clam.py:248-250): the 128 products of branch k sit in waves 2 k and 2 k + 1
    float prod = k < NB ? m * p.wcls[k * S1 + col] : 0.f;
    prod = wave_sum_dpp(prod);
    if (lane == 0) sh[w] = prod;
    __syncthreads();
    if (tid < NB) p.logits[tid] = (sh[2 * tid] + sh[2 * tid + 1]) + p.bcls[tid];
}

template <int KS, int NB>
int launch_mb(const hipt_clam_weights* w, const void* bag, int N, int passes, float* A_raw, void* h1_img, float* partials, unsigned* ticket,
              float* M, float* logits, hipStream_t st) {
    constexpr int lds = image_bytes(KS) + 4 * TB_BYTES;
    auto k = abmil32_kernel<KS, NB>;
    auto kp = clam_mb_pool_kernel<NB>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(abmil32 / multi-branch) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("abmil32: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    Abmil32Params p;
    memset(&p, 0, sizeof(p));
    p.bag = (const bf16_t*)bag;
    p.N = N;
    p.nblocks = (N + 31) / 32;
    const int maxg = once.ncu[dev] < 256 ? once.ncu[dev] : 256;
    const int rounds = (p.nblocks + 4 * maxg - 1) / (4 * maxg);
    const int grid = ((p.nblocks + rounds - 1) / rounds + 3) / 4;
    p.nwaves = grid * 4;
    p.image = (const char*)w->stream_pk;
    p.bc = w->bc;
    p.A_raw = A_raw;
    p.h1_img = h1_img;
    p.attention_only = 1;
    if (passes & 1) {  // passes: bit 0 = the streaming pass (A_raw, h1 image), bit 1 = the pooling pass (M, logits) -- capi.hip books them apart
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, p);
        HIPT_CHECK_LAUNCH();
    }
    if (!(passes & 2)) return HIPT_OK;
    ClamMbPoolParams q;
    memset(&q, 0, sizeof(q));
    q.h1_img = (const char*)h1_img;
    q.A_raw = A_raw;
    q.bc = w->bc;
    q.N = N;
    q.nblocks = p.nblocks;
    q.partials = partials;
    q.ticket = ticket;
    q.wcls = w->wcls;
    q.bcls = w->bcls;
    q.M = M;
    q.logits = logits;
    // (the partials of at most 128 workgroups -- hipt_clam_mb_workspace_bytes; a workgroup of eight waves walks at least eight blocks)
    int gp = (p.nblocks + 7) / 8;
    gp = gp < 1 ? 1 : (gp > 128 ? 128 : gp);
    hipLaunchKernelGGL(kp, dim3(gp), dim3(512), 0, st, q);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

}  // namespace

// CLAM_MB's one-pass inference (K = w->n_att attention branches, 2 <= K <= 4) where the streaming kernel runs: bf16 [384 | 192, 128, 64], the
// image packed WITH its K rows of wc, the largest of the K logit bounds below 60
bool hipt_clam_mb_stream_supported(const hipt_clam_weights* w) {
    return w->n_att >= 2 && w->n_att <= 4 && w->n_classes == w->n_att && hipt_clam_stream_supported(w);
}

size_t hipt_clam_mb_h1_bytes(int N) { return (size_t)((N + 31) / 32) * 8192; }

int hipt_clam_mb_stream_launch(const hipt_clam_weights* w, const void* bag, int N, int passes, float* A_raw, void* h1_img, float* partials,
                               unsigned* ticket, float* M, float* logits, hipStream_t st) {
    HIPT_CHECK_ARG((int64_t)N * w->s0 * 2 < (int64_t)0x7fff0000 && (int64_t)((N + 31) / 32) * 8192 < (int64_t)0x7fff0000, "clam stream: bag beyond 2 GiB");
#define MB_CASE(KS_, NB_) \
    if (w->s0 == 16 * KS_ && w->n_att == NB_) return launch_mb<KS_, NB_>(w, bag, N, passes, A_raw, h1_img, partials, ticket, M, logits, st);
    MB_CASE(24, 2) MB_CASE(24, 3) MB_CASE(24, 4) MB_CASE(12, 2) MB_CASE(12, 3) MB_CASE(12, 4)
#undef MB_CASE
    hipt_set_error("clam stream (multi-branch): unsupported S0=%d / branches=%d", w->s0, w->n_att);
    return HIPT_E_UNSUPPORTED;
}

// bf16 [384 | 192, 128, 64] with a usable logit bound: e^B times the row count times |h1| must stay inside fp32's range
bool hipt_clam_stream_supported(const hipt_clam_weights* w) {
    return w->dtype == HIPT_BF16 && w->s1 == S1 && w->s2 == S2 && (w->s0 == 384 || w->s0 == 192) && w->stream_pk && w->logit_bound > 0.f &&
           w->logit_bound < 60.f && !hipt_generic_only();
}

size_t hipt_clam_stream_image_bytes(const hipt_clam_weights* w) {
    if (w->dtype != HIPT_BF16 || w->s1 != S1 || w->s2 != S2 || (w->s0 != 384 && w->s0 != 192)) return 0;
    return (size_t)image_bytes(w->s0 / 16);
}

int hipt_clam_stream_pack_launch(const hipt_clam_weights* w, void* out, hipStream_t st) {
    const size_t nb = hipt_clam_stream_image_bytes(w);
    if (!nb) {
        hipt_set_error("clam stream pack: no image for dtype %d [%d,%d,%d]", w->dtype, w->s0, w->s1, w->s2);
        return HIPT_E_UNSUPPORTED;
    }
    HIPT_CHECK_ARG(w->w1 && w->b1 && w->wab && w->bab && w->wc && out && ((uintptr_t)out & 15) == 0, "clam stream pack: null / unaligned pointer");
    const int n = (int)(nb / 16);
    const int nbr = w->n_att > 1 ? w->n_att : 1;  // (CLAM_MB: wc = [n_att][S2])
    HIPT_CHECK_ARG(nbr <= 16, "clam stream pack: at most 16 attention branches (got %d)", nbr);
    hipLaunchKernelGGL(abmil32_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const bf16_t*)w->w1, w->b1, (const bf16_t*)w->wab, w->bab,
                       w->wc, w->s0 / 16, (char*)out, nbr);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_clam_stream_launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* partials, int* n_partials,
                            unsigned* ticket, float* M, float* logits, float* Y_prob, int64_t* Y_hat, hipStream_t st) {
    HIPT_CHECK_ARG((int64_t)N * w->s0 * 2 < (int64_t)0x7fff0000, "clam stream: bag beyond 2 GiB");
    if (w->s0 == 384) return launch<24>(w, bag, N, attention_only, A_raw, partials, n_partials, ticket, M, logits, Y_prob, Y_hat, st);
    if (w->s0 == 192) return launch<12>(w, bag, N, attention_only, A_raw, partials, n_partials, ticket, M, logits, Y_prob, Y_hat, st);
    hipt_set_error("clam stream: unsupported S0=%d", w->s0);
    return HIPT_E_UNSUPPORTED;
}
