// Fused softmax attention, bf16, head dim 64, up to 288 tokens, no probability output: the ViT-256 hot case of
// Attention.forward (HIPT_4K/vision_transformer.py:119-128): S = q k^T * scale, row softmax, O = P v.
//
// Same algorithm and LDS images as attention.hip (one (sequence, head) per 4-wave workgroup, two workgroups per
// CU; scores transposed so that a lane owns 4 consecutive keys of one query per 16-key tile; the exponentiated
// registers are the operand of the second product, V through the transposing LDS read).  What changes:
//   * K and V are staged by LDS-DMA (global_load_lds, 16 B per lane, the bank swizzle applied on the SOURCE address):
//     no VGPR round trip, the whole 72 KB in flight at once.  Rows past ntok are clamped to the last token (their
//     keys are masked, their probabilities are 0): nothing to zero.
//   * a wave takes its 16-query tiles two at a time: every K fragment and every transposed V fragment read from
//     LDS feeds both tiles (the kernel was LDS-issue and VALU bound, not MFMA bound).
//   * softmax in the exp2 domain with the scale folded into one FMA per element (v_exp_f32 directly: the inputs are
//     <= 0, no range handling needed), max3, masks only on the last two key tiles (the kernel takes 256 < ntok <= 288;
//     other lengths stay on attention.hip).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int DH = 64, NKT = 18, ROWS = NKT * 16, RB = 128;  // 288 key rows of 128 bytes

// K image: 16-byte chunk c of row r at physical chunk c ^ ((r >> 1) & 7) (16 consecutive rows read at one logical
// chunk cover 16 distinct 16-byte slots of two 256-byte bank rows)
__device__ __forceinline__ int k_off(int row, int c) { return row * RB + ((c ^ ((row >> 1) & 7)) << 4); }
// V image: 32-byte segment s of row r at s ^ ((r >> 1) & 3)
__device__ __forceinline__ int v_off(int row, int seg) { return row * RB + ((seg ^ ((row >> 1) & 3)) << 5); }

// Q fragments of one 16-query tile straight from global: row q, chunks g and g + 4
__device__ __forceinline__ void load_q(const bf16_t* __restrict__ qbase, int64_t tokstride, int ntok, int qt, int li, int g, u32x4 (&qf)[2]) {
    const int q = qt * 16 + li;
    const int qc = q < ntok ? q : ntok - 1;
    qf[0] = *(const u32x4*)(qbase + qc * tokstride + g * 8);
    qf[1] = *(const u32x4*)(qbase + qc * tokstride + (g + 4) * 8);
}

// S^T tiles: s[a][t][i] = <k[16t + 4g + i], q_a[li]>;  NQ = query tiles processed together (2, or 1 for an odd one)
// NKS: key tiles that hold tokens (17 for 257 tokens: the 18th tile's scores are neither computed nor exponentiated)
template <int NQ, int NKS>
__device__ __forceinline__ void scores(const u32x4 (&qf)[2][2], const char* Ks, int li, int g, f32x4 (&s)[NQ][NKT]) {
#pragma unroll
    for (int t = 0; t < NKS; ++t) {
        const u32x4 k0 = *(const u32x4*)(Ks + k_off(t * 16 + li, g));
        const u32x4 k1 = *(const u32x4*)(Ks + k_off(t * 16 + li, g + 4));
#pragma unroll
        for (int a = 0; a < NQ; ++a) {
            s[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            Tr<bf16_t>::mma16(s[a][t], k0, qf[a][0]);
            Tr<bf16_t>::mma16(s[a][t], k1, qf[a][1]);
        }
        if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads from all being hoisted (registers)
    }
}

// softmax over the keys, P V, store
// oimg >= 0: the output is written as a bf16 activation image (kernels.h; D = 384): oimg = first row (b * ntok) of this
// sequence, obase = out + h * DH.  Row r, columns h*64 + 16 dt + 4g ..+3 = chunk 8h + 2dt + (g >> 1), half (g & 1).
template <int NQ, int NKS>
__device__ __forceinline__ void finish(f32x4 (&s)[NQ][NKT], bf16_t* __restrict__ obase, const char* Vs, int D, int ntok, float sl2e, int qt0,
                                       int qt1, int li, int g, int64_t oimg, int h) {
    // ---- softmax over keys, fp32, exp2 domain ----
    float inv[NQ];
#pragma unroll
    for (int a = 0; a < NQ; ++a) {
        // keys past the end can only sit in the last two key tiles that hold tokens (16 (NKS - 2) < ntok <= 16 NKS)
#pragma unroll
        for (int t = NKS - 2; t < NKS; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (t * 16 + 4 * g + i >= ntok) s[a][t][i] = -INFINITY;
        float m = fmaxf(fmaxf(s[a][0][0], s[a][0][1]), fmaxf(s[a][0][2], s[a][0][3]));
#pragma unroll
        for (int t = 1; t < NKS; ++t) {
            m = __builtin_fmaxf(__builtin_fmaxf(m, s[a][t][0]), s[a][t][1]);  // -> v_max3_f32
            m = __builtin_fmaxf(__builtin_fmaxf(m, s[a][t][2]), s[a][t][3]);
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float ms = -m * sl2e;
        const f32x2 sc2 = {sl2e, sl2e}, ms2 = {ms, ms};
        f32x2 l2 = {0.f, 0.f};
#pragma unroll
        for (int t = NKS; t < NKT; ++t) s[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};  // (token-free tiles: probability 0 in the P V pairs)
#pragma unroll
        for (int t = 0; t < NKS; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2 x = __builtin_elementwise_fma(f32x2{s[a][t][2 * h], s[a][t][2 * h + 1]}, sc2, ms2);  // v_pk_fma_f32
                f32x2 e;
                e[0] = __builtin_amdgcn_exp2f(x[0]);
                e[1] = __builtin_amdgcn_exp2f(x[1]);
                l2 += e;
                s[a][t][2 * h] = e[0];
                s[a][t][2 * h + 1] = e[1];
            }
        }
        float l = l2[0] + l2[1];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        inv[a] = 1.0f / l;
    }
    // ---- O^T = V^T P^T: o[a][dt][i] = O[q = li][d = 16dt + 4g + i]; key tiles in pairs (K = 32 per MFMA) ----
    f32x4 o[NQ][DH / 16];
#pragma unroll
    for (int a = 0; a < NQ; ++a)
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) o[a][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < NKT / 2; ++sp) {
        u32x4 pf[NQ];  // K slots j<4: key 32sp + 4g + j ; j>=4: key 32sp + 16 + 4g + (j-4)
#pragma unroll
        for (int a = 0; a < NQ; ++a) {
            pf[a][0] = pack_bf16x2(s[a][2 * sp][0], s[a][2 * sp][1]);
            pf[a][1] = pack_bf16x2(s[a][2 * sp][2], s[a][2 * sp][3]);
            pf[a][2] = pack_bf16x2(s[a][2 * sp + 1][0], s[a][2 * sp + 1][1]);
            pf[a][3] = pack_bf16x2(s[a][2 * sp + 1][2], s[a][2 * sp + 1][3]);
        }
        // transposing read: lane (g, li) supplies row 4g + (li>>2) of the 4x16 block, columns 4*(li&3)..
        const int r0 = 32 * sp + 4 * g + (li >> 2);
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vs + v_off(r0, dt) + ((li & 3) << 3)));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vs + v_off(r0 + 16, dt) + ((li & 3) << 3)));
            u32x4 vf;
            const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
            vf[0] = lo2[0];
            vf[1] = lo2[1];
            vf[2] = hi2[0];
            vf[3] = hi2[1];
#pragma unroll
            for (int a = 0; a < NQ; ++a) Tr<bf16_t>::mma16(o[a][dt], vf, pf[a]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int a = 0; a < NQ; ++a) {
        const int q = (a == 0 ? qt0 : qt1) * 16 + li;
        if (q < ntok) {
            if (oimg >= 0) {
                const int64_t r = oimg + q;
                bf16_t* orow = obase + (r >> 4) * (16 * 384) + (2 * h) * 512 + ((int)(r & 15)) * 8 + (g >> 1) * 128 + 4 * (g & 1);
#pragma unroll
                for (int dt = 0; dt < DH / 16; ++dt)  // chunk 8h + 2dt + (g>>1): c = 2h + (dt >> 1), lane group 2(dt & 1) + (g >> 1)
                    store4<bf16_t>(orow + (dt >> 1) * 512 + (dt & 1) * 256, o[a][dt] * inv[a]);
            } else {
                bf16_t* orow = obase + (int64_t)q * D + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DH / 16; ++dt) store4<bf16_t>(orow + dt * 16, o[a][dt] * inv[a]);
            }
        }
    }
}

template <int NKS>
__global__ __launch_bounds__(256, 2) void attn64_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int ntok, int heads, float sl2e,
                                                         int out_img, int qkv_hm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + ROWS * RB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int D = heads * DH;
    // qkv row-major [b, token, (q|k|v) x head x 64], or head-major [b][q/k/v][head][token][64] (seqgemm_pipe, OHM)
    const int64_t tokstride = qkv_hm ? DH : 3 * (int64_t)D;
    const int64_t mstride = qkv_hm ? (int64_t)heads * ntok * DH : D;  // q -> k -> v
    const bf16_t* qbase = qkv_hm ? qkv + ((int64_t)b * 3 * heads + h) * ntok * DH : qkv + (int64_t)b * ntok * tokstride + h * DH;
    const bf16_t* kbase = qbase + mstride;
    const bf16_t* vbase = qbase + 2 * mstride;

    const int g = lane >> 4, li = lane & 15;
    const int nqt = (ntok + 15) >> 4;
    // this wave's tiles: wave, wave+4, ...; two at a time.  The first pair's Q rows are requested before the staging
    // wait; every later step's Q is requested right after the step before it has turned its Q into scores (the
    // registers are free then, and the softmax / PV that follow hide the latency).
    // (257 tokens are 17 tiles: one wave of the four has five.  Which one rotates with the workgroup, so that the two workgroups
    //  sharing a CU do not both load the same SIMD.)
    // (gridDim.y > 1, calls of a few sequences: the query tiles are also dealt over gridDim.y workgroups, each staging K and V for
    //  itself -- 6 workgroups would leave 250 CUs idle for 12 us; a query row's arithmetic is the same wherever it runs)
    const int QS = 4 * gridDim.y;
    int qt = ((wave + blockIdx.x) & 3) + 4 * blockIdx.y;
    u32x4 qf[2][2];
    if (qt < nqt) load_q(qbase, tokstride, ntok, qt, li, g, qf[0]);
    if (qt + QS < nqt) load_q(qbase, tokstride, ntok, qt + QS, li, g, qf[1]);

    // ---- stage K and V by LDS-DMA: piece = 8 LDS rows (1 KiB); lane i fills row 8 piece + i/8, physical chunk i%8 ----
    {
        const int rr = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int pi = 0; pi < ROWS / 8 / 4; ++pi) {  // 36 pieces per matrix, 9 per wave
            const int piece = pi * 4 + wave;
            const int r = piece * 8 + rr;
            const int rc = r < ntok ? r : ntok - 1;
            const int kc = pc ^ ((r >> 1) & 7);                              // logical chunk stored at this K slot
            const int vc = (((pc >> 1) ^ ((r >> 1) & 3)) << 1) | (pc & 1);  // ... at this V slot
            glds16(kbase + rc * tokstride + kc * 8, Ks + piece * 1024);
            glds16(vbase + rc * tokstride + vc * 8, Vs + piece * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    bf16_t* obase = out_img ? out : out + (int64_t)b * ntok * D + h * DH;
    const int64_t oimg = out_img ? (int64_t)b * ntok : -1;
    for (; qt + QS < nqt; qt += 2 * QS) {
        f32x4 s[2][NKT];
        scores<2, NKS>(qf, Ks, li, g, s);
        if (qt + 2 * QS < nqt) load_q(qbase, tokstride, ntok, qt + 2 * QS, li, g, qf[0]);
        if (qt + 3 * QS < nqt) load_q(qbase, tokstride, ntok, qt + 3 * QS, li, g, qf[1]);
        finish<2, NKS>(s, obase, Vs, D, ntok, sl2e, qt, qt + QS, li, g, oimg, h);
    }
    if (qt < nqt) {
        f32x4 s[1][NKT];
        scores<1, NKS>(qf, Ks, li, g, s);
        finish<1, NKS>(s, obase, Vs, D, ntok, sl2e, qt, qt, li, g, oimg, h);
    }
}

}  // namespace

bool hipt_attention64_supported(int dtype, int dh, int ntok, bool want_probs) {
    return dtype == HIPT_BF16 && dh == 64 && ntok > ROWS - 32 && ntok <= ROWS && !want_probs;
}

int hipt_attention64_launch(const void* qkv, void* out, int B, int ntok, int heads, float scale, hipStream_t st, int out_img, int qkv_hm) {
    HIPT_CHECK_ARG(!out_img || (heads * DH == 384 && ((int64_t)B * ntok) % 16 == 0), "attention64: image output needs D = 384 and whole 16-row fragments");
    constexpr int lds = 2 * ROWS * RB;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)attn64_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)attn64_kernel<NKT - 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(attention64) failed");
            return HIPT_E_LAUNCH;
        }
        once.done[dev] = true;
    }
    const int qsplit = B * heads <= 48 ? (((ntok + 15) >> 4) + 3) / 4 : 1;  // few sequences: one query tile per wave
    hipLaunchKernelGGL(ntok <= ROWS - 16 ? attn64_kernel<NKT - 1> : attn64_kernel<NKT>, dim3(B * heads, qsplit), dim3(256), lds, st, (const bf16_t*)qkv, (bf16_t*)out, ntok, heads,
                       scale * 1.4426950408889634f, out_img, qkv_hm);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}
