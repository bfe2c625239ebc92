// Fused softmax attention for one (sequence, head) per workgroup — the core of Attention.forward
// (HIPT_4K/vision_transformer.py:119-128): S = q k^T * scale, row softmax, O = P v.  The S matrix
// ([B,6,257,257] fp32 = 406 MB per block in the reference) never leaves the chip.
//
// gfx950 design: all keys/values of the head (<= 288 tokens) are staged once into LDS
// (bank-conflict-free XOR swizzles, see below); each wave owns 16-query tiles.  Scores are computed
// TRANSPOSED (S^T = K Q^T, keys on the MFMA row axis) so that after the MFMA every lane holds, for
// ONE query (lane & 15), 4 consecutive keys per 16-key tile: the row max / sum are lane-local plus
// two shuffles, and the exponentiated registers are — unchanged, no LDS round trip — the operand
// fragment of the second product O^T = V^T P^T (accumulator-as-operand, K order permuted
// consistently on the V side).  V fragments come from row-major V through the hardware
// transposing LDS read (ds_read_b64_tr_b16) in bf16 mode and through padded ds_read_b32 in fp32
// mode.  Softmax runs in fp32 with exp2 and a pre-multiplied scale*log2(e).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

template <typename T, int DH> struct Geo {
    static constexpr int ESZ = sizeof(T);
    static constexpr int RB = DH * ESZ;         // K row bytes: 64 / 128 / 256
    static constexpr int CPR = RB / 16;         // 16-byte chunks per row: 4 / 8 / 16
    static constexpr int RPB = 256 / RB < 1 ? 1 : 256 / RB;  // rows per 256-byte bank row
    static constexpr int KS = RB / 64;          // MFMA K steps per row (4 lane groups x 16 B)
    // V: bf16 rows are DH*2 bytes, swizzled per 32-byte segment; fp32 rows are padded to DH+4 floats
    static constexpr int VRB = ESZ == 2 ? DH * 2 : (DH + 4) * 4;
    static constexpr int SPR = DH / 16;         // 32-byte segments per bf16 V row: 2 / 4
    static constexpr int VRPB = 256 / (DH * 2); // bf16 V rows per bank row: 4 / 2
};

// K image: chunk c of row r lives at physical chunk c ^ ((r / RPB) & (CPR-1)): 16 consecutive rows
// read at the same logical chunk (one ds_read_b128 lane group) then cover 16 distinct 16-B slots.
template <typename T, int DH> __device__ __forceinline__ int k_off(int row, int c) {
    using G = Geo<T, DH>;
    return row * G::RB + ((c ^ ((row / G::RPB) & (G::CPR - 1))) << 4);
}
// bf16 V image: 32-byte segment s of row r lives at s ^ ((r / VRPB) & (SPR-1)): the 8 rows a
// 32-lane half touches in one ds_read_b64_tr_b16 then hit 8 distinct 32-byte slots of the bank row.
template <int DH> __device__ __forceinline__ int v_off_bf16(int row, int seg) {
    using G = Geo<bf16_t, DH>;
    return row * G::VRB + ((seg ^ ((row / G::VRPB) & (G::SPR - 1))) << 5);
}

template <typename T, int DH, int NKT, bool WRITE_P>
__global__ __launch_bounds__(256, 2) void attn_kernel(const T* __restrict__ qkv, T* __restrict__ out,
                                                      float* __restrict__ probs, int ntok, int heads, float sl2e) {
    using G = Geo<T, DH>;
    constexpr int ROWS = NKT * 16;
    constexpr int EPC = Tr<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + ROWS * G::RB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int D = heads * DH;
    const int64_t tokstride = 3 * (int64_t)D;
    const T* qbase = qkv + (int64_t)b * ntok * tokstride + h * DH;
    const T* kbase = qbase + D;
    const T* vbase = qbase + 2 * D;

    // ---- stage K and V (zero rows beyond ntok so that masked P (=0) times V stays 0) ----
    for (int idx = tid; idx < ROWS * G::CPR; idx += 256) {
        const int row = idx / G::CPR, c = idx % G::CPR;
        u32x4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
        if (row < ntok) {
            kv = *(const u32x4*)(kbase + row * tokstride + c * EPC);
            vv = *(const u32x4*)(vbase + row * tokstride + c * EPC);
        }
        *(u32x4*)(Ks + k_off<T, DH>(row, c)) = kv;
        if constexpr (sizeof(T) == 2)
            *(u32x4*)(Vs + v_off_bf16<DH>(row, c >> 1) + ((c & 1) << 4)) = vv;
        else
            *(u32x4*)(Vs + row * G::VRB + (c << 4)) = vv;
    }
    __syncthreads();

    const int g = lane >> 4, li = lane & 15;
    const int nqt = (ntok + 15) >> 4;
    // (gridDim.y > 1, calls of a few sequences: the query tiles are dealt over gridDim.y workgroups as well, each staging K and V
    //  for itself -- six workgroups per patch would leave 250 CUs idle; a query row's arithmetic is the same wherever it runs)
    for (int qt = wave + 4 * blockIdx.y; qt < nqt; qt += 4 * gridDim.y) {
        // ---- Q fragments straight from global: row q, chunks g + 4*ks ----
        int q = qt * 16 + li;
        const bool qvalid = q < ntok;
        const int qc = qvalid ? q : ntok - 1;
        u32x4 qf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qf[ks] = *(const u32x4*)(qbase + qc * tokstride + (g + 4 * ks) * EPC);

        // ---- S^T tiles: s[t][i] = <k[16t + 4g + i], q[li]> ----
        f32x4 s[NKT];
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const u32x4 kf = *(const u32x4*)(Ks + k_off<T, DH>(t * 16 + li, g + 4 * ks));
                Tr<T>::mma16(s[t], kf, qf[ks]);
            }
            // keep the scheduler from hoisting every tile's LDS reads to the top (register blow-up)
            if (t & 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- softmax over keys (fp32): mask, max, exp2, sum ----
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (t * 16 + 4 * g + i >= ntok) s[t][i] = -INFINITY;
                m = fmaxf(m, s[t][i]);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[t][i] = exp2f((s[t][i] - m) * sl2e);
                l += s[t][i];
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;

        if constexpr (WRITE_P) {
            if (qvalid) {
                float* pr = probs + (((int64_t)b * heads + h) * ntok + q) * ntok;
#pragma unroll
                for (int t = 0; t < NKT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int key = t * 16 + 4 * g + i;
                        if (key < ntok) pr[key] = s[t][i] * inv;
                    }
            }
        }

        // ---- O^T = V^T P^T: o[dt][i] = O[q = li][d = 16dt + 4g + i] ----
        f32x4 o[DH / 16];
#pragma unroll
        for (int dt = 0; dt < DH / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (sizeof(T) == 2) {
            static_assert(sizeof(T) != 2 || NKT % 2 == 0, "bf16 PV consumes key tiles in pairs");
#pragma unroll
            for (int sp = 0; sp < NKT / 2; ++sp) {
                u32x4 pf;  // K slots j<4: key 32sp + 4g + j ; j>=4: key 32sp + 16 + 4g + (j-4)
                pf[0] = pack_bf16x2(s[2 * sp][0], s[2 * sp][1]);
                pf[1] = pack_bf16x2(s[2 * sp][2], s[2 * sp][3]);
                pf[2] = pack_bf16x2(s[2 * sp + 1][0], s[2 * sp + 1][1]);
                pf[3] = pack_bf16x2(s[2 * sp + 1][2], s[2 * sp + 1][3]);
                // transposing read: lane (g, li) supplies row 4g + (li>>2) of the 4x16 block, columns 4*(li&3)..
                const int r0 = 32 * sp + 4 * g + (li >> 2);
#pragma unroll
                for (int dt = 0; dt < DH / 16; ++dt) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (LDS_AS s16x4*)(Vs + v_off_bf16<DH>(r0, dt) + ((li & 3) << 3)));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (LDS_AS s16x4*)(Vs + v_off_bf16<DH>(r0 + 16, dt) + ((li & 3) << 3)));
                    u32x4 vf;
                    const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
                    vf[0] = lo2[0];
                    vf[1] = lo2[1];
                    vf[2] = hi2[0];
                    vf[3] = hi2[1];
                    Tr<T>::mma16(o[dt], vf, pf);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const u32x4 pf = __builtin_bit_cast(u32x4, s[t]);  // slot j: key 16t + 4g + j
                const float* vrow = (const float*)(Vs + (t * 16 + 4 * g) * G::VRB) + li;
#pragma unroll
                for (int dt = 0; dt < DH / 16; ++dt) {
                    u32x4 vf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) vf[j] = __builtin_bit_cast(uint32_t, vrow[j * (G::VRB / 4) + dt * 16]);
                    Tr<T>::mma16(o[dt], vf, pf);
                }
                if (t & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (qvalid) {
            T* orow = out + ((int64_t)b * ntok + q) * D + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < DH / 16; ++dt) store4<T>(orow + dt * 16, o[dt] * inv);
        }
    }
}

template <typename T, int DH, int NKT>
int launch(const void* qkv, void* out, float* probs, int B, int ntok, int heads, float scale, hipStream_t st) {
    using G = Geo<T, DH>;
    constexpr int lds = NKT * 16 * (G::RB + G::VRB);
    const float sl2e = scale * 1.4426950408889634f;
    const dim3 grid(B * heads, B * heads <= 48 ? (((ntok + 15) >> 4) + 3) / 4 : 1), block(256);  // few sequences: one query tile per wave
    if (probs) {
        auto k = attn_kernel<T, DH, NKT, true>;
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k, grid, block, lds, st, (const T*)qkv, (T*)out, probs, ntok, heads, sl2e);
    } else {
        auto k = attn_kernel<T, DH, NKT, false>;
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k, grid, block, lds, st, (const T*)qkv, (T*)out, probs, ntok, heads, sl2e);
    }
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

template <typename T>
int dispatch(const void* qkv, void* out, float* probs, int B, int ntok, int heads, int dh, float scale, hipStream_t st) {
    if (dh == 64) {
        if (ntok <= 32) return launch<T, 64, 2>(qkv, out, probs, B, ntok, heads, scale, st);
        return launch<T, 64, 18>(qkv, out, probs, B, ntok, heads, scale, st);
    }
    if (ntok <= 32) return launch<T, 32, 2>(qkv, out, probs, B, ntok, heads, scale, st);
    return launch<T, 32, 18>(qkv, out, probs, B, ntok, heads, scale, st);
}

}  // namespace

int hipt_attention_launch(const void* qkv, void* out, float* probs, int B, int ntok, int heads, int dh, float scale,
                          int dtype, hipStream_t st, int out_img, int qkv_hm) {
    HIPT_CHECK_ARG(B > 0 && heads > 0 && ntok > 0, "attention: empty problem");
    HIPT_CHECK_ARG(dh == 32 || dh == 64, "attention: head dim %d not in {32, 64}", dh);
    if (ntok > 288) {
        hipt_set_error("attention: ntok=%d exceeds the on-chip envelope (288 tokens)", ntok);
        return HIPT_E_UNSUPPORTED;
    }
    HIPT_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "attention: 16-byte alignment required");
    if (!hipt_generic_only() && hipt_attention64_supported(dtype, dh, ntok, probs != nullptr)) return hipt_attention64_launch(qkv, out, B, ntok, heads, scale, st, out_img, qkv_hm);
    HIPT_CHECK_ARG(!out_img && !qkv_hm, "attention: only the 64-wide-head bf16 kernel handles activation images / head-major qkv");
    if (dtype == HIPT_F32) return dispatch<float>(qkv, out, probs, B, ntok, heads, dh, scale, st);
    if (dtype == HIPT_BF16) return dispatch<bf16_t>(qkv, out, probs, B, ntok, heads, dh, scale, st);
    hipt_set_error("attention: bad dtype %d", dtype);
    return HIPT_E_BADARG;
}
