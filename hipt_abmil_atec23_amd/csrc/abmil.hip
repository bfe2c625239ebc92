// CLAM_SB / ABMIL gated-attention pooling (models/model_clam.py:41-64, 83-92, 147-183) as ONE
// streaming pass over the bag:
//   h1 = ReLU(bag W1^T + b1)            [N,S1]   (attention_net.0, :83)
//   A  = (tanh(h1 Wa^T + ba) * sigmoid(h1 Wb^T + bb)) wc + bc   [N]   (Attn_Net_Gated.forward :59-64)
//   M  = softmax_N(A) h1                [S1]     (:154,180)
// The reference issues 3 Linear + tanh + sigmoid + mul + Linear + transpose + softmax + mm as separate
// ops, each round-tripping [N,*] tensors through memory; here the bag is read from HBM exactly once
// and nothing of size N except A_raw is written.
//
// gfx950 design (fused kernel): persistent workgroups walk 128-row tiles of the bag.  Per tile:
//   phase 1  128 x S1 GEMM tile, K = S0 streamed in 128-byte slabs by LDS-DMA (2-stage ring, same
//            swizzled image as gemm.hip); accumulators -> +b1, ReLU -> LDS as the A-operand image
//            of phase 2 (h1 never reaches HBM);
//   phase 2  128 x 2*S2 GEMM tile against [Wa;Wb] (rows interleaved a,a,b,b per 4 so that one lane
//            owns a_j and b_j of the same j), tanh * sigmoid * wc summed in registers, 2 shuffles,
//            one LDS hop across the two column waves -> A_raw[128];
//   pooling  online softmax over tiles (running max / sum per workgroup) and the weighted sum
//            p^T h1 as MFMAs on the h1 image (transposing LDS read in bf16 mode).
// Each workgroup leaves (max, sum, acc[S1]); a one-workgroup combine kernel merges them and applies
// the bag classifier, softmax and argmax.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int TM = 128;  // rows per tile

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
    // 1 - 2/(e^{2x}+1): exact limits at +-inf, abs error ~1e-7 elsewhere
    return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f);
}

template <typename T, int S1, int S2> struct AG {
    static constexpr int KB = Tr<T>::KB;
    static constexpr int NSLAB = S1 / KB;             // 128-byte slabs of the h1 image
    static constexpr int NJ1 = S1 / 32;               // n-frags per wave, phase 1 (2 column waves)
    static constexpr int NJ2 = (2 * S2) / 32;         // n-frags per wave, phase 2
    static constexpr int STAGE = (TM + S1) * 128;
    static constexpr int H1_BYTES = NSLAB * TM * 128;
    static constexpr int WAB_BYTES = NSLAB * 2 * S2 * 128;
    static constexpr int AREA = (2 * STAGE > H1_BYTES + WAB_BYTES) ? 2 * STAGE : H1_BYTES + WAB_BYTES;
    static constexpr int LDS = AREA + TM * 4 * 3 + 64;  // + A_raw[128], partial[2][128], scalars
    static_assert(S1 % KB == 0 && S1 % 32 == 0 && S1 <= 128, "fused ABMIL: S1 in {32(bf16: 64),64,128}");
    static_assert((2 * S2) % 32 == 0 && 2 * S2 <= 128, "fused ABMIL: S2 in {16,32,64}");
};

// byte offset of element (row, col) inside the slab-major, swizzled A-operand image of h1
template <typename T> __device__ __forceinline__ int h1_off(int row, int col) {
    constexpr int KB = Tr<T>::KB, EPC = Tr<T>::EPC;
    const int slab = col / KB, c = (col % KB) / EPC, sub = (col % EPC) * (int)sizeof(T);
    return slab * (TM * 128) + row * 128 + ((c ^ ((row >> 1) & 7)) << 4) + sub;
}

template <typename T, int S1, int S2>
__global__ __launch_bounds__(256, 2) void abmil_fused_kernel(const T* __restrict__ bag, int N, int S0,
                                                             const T* __restrict__ w1, const float* __restrict__ b1,
                                                             const T* __restrict__ wab, const float* __restrict__ bab,
                                                             const float* __restrict__ wc, const float* __restrict__ bc,
                                                             float* __restrict__ A_raw, float* __restrict__ partials,
                                                             int attention_only) {
    using G = AG<T, S1, S2>;
    constexpr int EPC = Tr<T>::EPC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* H1s = smem;                     // aliases the stage ring (used after phase 1)
    char* Wabs = smem + G::H1_BYTES;
    float* As = (float*)(smem + G::AREA);        // A_raw of the tile
    float* Ps = As + TM;                          // [2][TM] per-column-wave partial gate sums
    float* Sc = Ps + 2 * TM;                      // scalars

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;
    const int ntiles = (N + TM - 1) / TM;
    const int nk = S0 / G::KB;

    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = li * 128 + (((g + 4 * ks) ^ ((lane >> 1) & 7)) << 4);

    // per-lane LDS-DMA geometry: row within an 8-row instruction block and logical chunk
    const int drow = lane >> 3;

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 accM[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};  // c-frags 2*wave, 2*wave+1

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        // ---------------- phase 1: h1pre = bag_tile @ W1^T ----------------
        const T* xsrc[4];
        int xch[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 8 + drow;
            int m = m0 + r;
            m = m < N ? m : N - 1;
            xsrc[q] = bag + (int64_t)m * S0;
            xch[q] = (lane & 7) ^ ((r >> 1) & 7);
        }
        auto stage = [&](int s, int kt) {
            char* sa = smem + s * G::STAGE;
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16(xsrc[q] + (kt * 8 + xch[q]) * EPC, sa + (wave * 4 + q) * 1024);
#pragma unroll
            for (int q = 0; q < S1 / 32; ++q) {  // S1 rows of W1: S1/8 instructions over 4 waves
                const int r = (wave * (S1 / 32) + q) * 8 + drow;
                glds16(w1 + (int64_t)r * S0 + (kt * 8 + ((lane & 7) ^ ((r >> 1) & 7))) * EPC,
                       sa + TM * 128 + (wave * (S1 / 32) + q) * 1024);
            }
        };
        f32x4 acc1[4][G::NJ1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < G::NJ1; ++j) acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        __syncthreads();  // previous tile's readers of the aliased area are done
        stage(0, 0);
        wait_vm0();
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const char* sa = smem + cur * G::STAGE + wm * 64 * 128;
            const char* sw = smem + cur * G::STAGE + TM * 128 + wn * (S1 / 2) * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 af[4], wf[G::NJ1];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const u32x4*)(sa + i * 16 * 128 + foff[ks]);
#pragma unroll
                for (int j = 0; j < G::NJ1; ++j) wf[j] = *(const u32x4*)(sw + j * 16 * 128 + foff[ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < G::NJ1; ++j) Tr<T>::mma16(acc1[i][j], wf[j], af[i]);
            }
            wait_vm0();
            __syncthreads();
            cur ^= 1;
        }
        // ---------------- [Wa;Wb] image by LDS-DMA (rows interleaved a,a,b,b) ----------------
        // packed row r: quad = r>>2, pos = r&3 -> source row (pos>>1)*S2 + quad*2 + (pos&1)
#pragma unroll
        for (int sl = 0; sl < G::NSLAB; ++sl)
#pragma unroll
            for (int q = 0; q < (2 * S2) / 32; ++q) {
                const int blk = wave * ((2 * S2) / 32) + q;
                const int r = blk * 8 + drow;
                const int srow = ((r & 3) >> 1) * S2 + (r >> 2) * 2 + (r & 1);
                glds16(wab + (int64_t)srow * S1 + (sl * 8 + ((lane & 7) ^ ((r >> 1) & 7))) * EPC,
                       Wabs + sl * (2 * S2 * 128) + blk * 1024);
            }
        // ---------------- h1 = ReLU(acc1 + b1) -> LDS image ----------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 64 + i * 16 + li;
#pragma unroll
            for (int j = 0; j < G::NJ1; ++j) {
                const int col = wn * (S1 / 2) + j * 16 + 4 * g;
                f32x4 v = acc1[i][j] + *(const f32x4*)(b1 + col);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                store4<T>((T*)(H1s + h1_off<T>(row, col)), v);
            }
        }
        wait_vm0();
        __syncthreads();
        // ---------------- phase 2: ab = h1 @ [Wa;Wb]^T, gate, reduce ----------------
        f32x4 acc2[4][G::NJ2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < G::NJ2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < G::NSLAB; ++sl) {
            const char* sa = H1s + sl * (TM * 128) + wm * 64 * 128;
            const char* sw = Wabs + sl * (2 * S2 * 128) + wn * S2 * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 af[4], wf[G::NJ2];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *(const u32x4*)(sa + i * 16 * 128 + foff[ks]);
#pragma unroll
                for (int j = 0; j < G::NJ2; ++j) wf[j] = *(const u32x4*)(sw + j * 16 * 128 + foff[ks]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < G::NJ2; ++j) Tr<T>::mma16(acc2[i][j], wf[j], af[i]);
            }
        }
        {
            float gate[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < G::NJ2; ++j) {
                const int r0 = wn * S2 + j * 16 + 4 * g;  // packed row of element 0
                const int j0 = (r0 >> 2) * 2;             // gate unit of elements 0 (a) and 2 (b); j0+1 for 1 and 3
                const float ba0 = bab[j0], ba1 = bab[j0 + 1], bb0 = bab[S2 + j0], bb1 = bab[S2 + j0 + 1];
                const float c0 = wc[j0], c1 = wc[j0 + 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x4 v = acc2[i][j];
                    gate[i] += tanh_f(v[0] + ba0) * sigmoid_f(v[2] + bb0) * c0 +
                               tanh_f(v[1] + ba1) * sigmoid_f(v[3] + bb1) * c1;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = gate[i];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (g == 0) Ps[wn * TM + wm * 64 + i * 16 + li] = v;
            }
        }
        __syncthreads();
        float a_mine = -INFINITY;  // threads 0..127 own one row each
        if (tid < TM) {
            const int m = m0 + tid;
            if (m < N) {
                a_mine = Ps[tid] + Ps[TM + tid] + bc[0];
                A_raw[m] = a_mine;
            }
            As[tid] = a_mine;
        }
        if (attention_only) continue;  // uniform
        // ---------------- pooling: online softmax + p^T h1 ----------------
        {
            float mt = wave_max(a_mine);
            if (lane == 0 && wave < 2) Sc[wave] = mt;
        }
        __syncthreads();
        const float m_new = fmaxf(m_run, fmaxf(Sc[0], Sc[1]));  // finite: every tile has >= 1 valid row
        const float resc = exp2f((m_run - m_new) * 1.4426950408889634f);  // 0 on the first tile
        m_run = m_new;
        // p for this lane's K slots (rows of the tile); invalid rows carry -inf -> p = 0
        float lsum = 0.f;
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int mb = 0; mb < TM; mb += 32) {
                float p[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p[e] = expf(As[mb + 4 * g + e] - m_new);
                    p[4 + e] = expf(As[mb + 16 + 4 * g + e] - m_new);
                }
                u32x4 pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pf[e] = pack_bf16x2(p[2 * e], p[2 * e + 1]);
                    // the sum must use the SAME rounded weights as the MFMA so that M is a true convex mix
                    const uint32_t packed = pf[e];  // scalar copy: never bit_cast a vector element in place
                    const bf16x2 r = __builtin_bit_cast(bf16x2, packed);
                    lsum += (float)r[0] + (float)r[1];
                }
                const int r0 = mb + 4 * g + (li >> 2);
#pragma unroll
                for (int cf = 0; cf < 2; ++cf) {
                    const int col = (wave * 2 + cf) * 16 + 4 * (li & 3);
                    if ((wave * 2 + cf) * 16 < S1) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(H1s + h1_off<T>(r0, col)));
                        const s16x4 hi =
                            __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(H1s + h1_off<T>(r0 + 16, col)));
                        const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
                        u32x4 hf;
                        hf[0] = lo2[0];
                        hf[1] = lo2[1];
                        hf[2] = hi2[0];
                        hf[3] = hi2[1];
                        Tr<T>::mma16(o[cf], hf, pf);
                    }
                }
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < TM; mb += 16) {
                u32x4 pf;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = expf(As[mb + 4 * g + e] - m_new);
                    lsum += p;
                    pf[e] = __builtin_bit_cast(uint32_t, p);
                }
#pragma unroll
                for (int cf = 0; cf < 2; ++cf) {
                    const int col = (wave * 2 + cf) * 16 + li;
                    if ((wave * 2 + cf) * 16 < S1) {
                        u32x4 hf;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            hf[e] = *(const uint32_t*)(H1s + h1_off<T>(mb + 4 * g + e, col));
                        Tr<T>::mma16(o[cf], hf, pf);
                    }
                }
            }
        }
        // lsum: lane (g, li) summed the rows of its K slots; the 4 groups partition the tile's rows and
        // all 16 li lanes of a group hold the same value -> total = sum over g = 2 shuffles
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        l_run = l_run * resc + lsum;
#pragma unroll
        for (int cf = 0; cf < 2; ++cf) accM[cf] = accM[cf] * resc + o[cf];
    }
    // ---------------- per-workgroup partial: (max, sum, acc[S1]) ----------------
    if (!attention_only) {
        float* pw = partials + (int64_t)blockIdx.x * (2 + S1);
        if (tid == 0) {
            pw[0] = m_run;
            pw[1] = l_run;
        }
        if (li == 0) {
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
                if ((wave * 2 + cf) * 16 < S1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) pw[2 + (wave * 2 + cf) * 16 + 4 * g + e] = accM[cf][e];
        }
    }
}

// Merge per-workgroup partials; bag classifier, softmax, argmax (model_clam.py:180-183).
// One 1024-thread workgroup: the G rescale factors exp(m_g - m*) are computed once into LDS, then
// thread (c, part) sums column c over every 8th partial (coalesced 4*S1-byte rows), LDS tree over parts.
constexpr int CMB_MAXG = 1024;
__global__ __launch_bounds__(1024) void abmil_combine_kernel(const float* __restrict__ partials, int G, int S1,
                                                             const float* __restrict__ wcls, const float* __restrict__ bcls,
                                                             int C, float* __restrict__ M, float* __restrict__ logits,
                                                             float* __restrict__ Y_prob, int64_t* __restrict__ Y_hat) {
    extern __shared__ float sm[];  // [CMB_MAXG] factors | [8*S1] column partial sums | [S1] M | [C] logits
    float* Fs = sm;
    float* Cs = Fs + CMB_MAXG;
    float* Ms = Cs + 8 * S1;
    float* Ls = Ms + S1;
    __shared__ float red[16];
    const int tid = threadIdx.x, stride = 2 + S1, wv = tid >> 6, ln = tid & 63;
    float mx = -INFINITY;
    for (int gi = tid; gi < G; gi += 1024) mx = fmaxf(mx, partials[(int64_t)gi * stride]);
    mx = wave_max(mx);
    if (ln == 0) red[wv] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, red[i]);
    __syncthreads();
    float ls = 0.f;
    for (int gi = tid; gi < G; gi += 1024) {
        const float f = expf(partials[(int64_t)gi * stride] - mx);
        Fs[gi] = f;
        ls += partials[(int64_t)gi * stride + 1] * f;
    }
    ls = wave_sum(ls);
    if (ln == 0) red[wv] = ls;
    __syncthreads();
    float L = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) L += red[i];
    // column sums: S1 <= 128 columns x 8 parts = 1024 threads (wider S1: loop)
    for (int c0 = 0; c0 < S1; c0 += 128) {
        const int c = c0 + (tid & 127), part = tid >> 7;
        float a = 0.f;
        if (c < S1)
            for (int gi = part; gi < G; gi += 8) a += partials[(int64_t)gi * stride + 2 + c] * Fs[gi];
        if (c < S1) Cs[part * S1 + c] = a;
    }
    __syncthreads();
    for (int c = tid; c < S1; c += 1024) {
        float a = 0.f;
#pragma unroll
        for (int part = 0; part < 8; ++part) a += Cs[part * S1 + c];
        a /= L;
        Ms[c] = a;
        M[c] = a;
    }
    __syncthreads();
    for (int k = wv; k < C; k += 16) {
        float a = 0.f;
        for (int c = ln; c < S1; c += 64) a += Ms[c] * wcls[(int64_t)k * S1 + c];
        a = wave_sum(a);
        if (ln == 0) Ls[k] = a + bcls[k];
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

// ---------------- generic (any width) building blocks ----------------
// A[m] = sum_j tanh(ab[m][j] ) * sigmoid(ab[m][S2 + j]) * wc[j] + bc   (biases already added by the GEMM)
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ ab, int64_t ld, int N, int S2,
                                                   const float* __restrict__ wc, const float* __restrict__ bc,
                                                   float* __restrict__ A) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    const float* r = ab + (int64_t)row * ld;
    float a = 0.f;
    for (int j = lane; j < S2; j += 64) a += tanh_f(r[j]) * sigmoid_f(r[S2 + j]) * wc[j];
    a = wave_sum(a);
    if (lane == 0) A[row] = a + bc[0];
}

// ab[m][j] = x[m] . wab[j] + bab[j] for widths the MFMA path does not cover (VALU; tiny K only)
__device__ __forceinline__ float ld_any(const void* p, int dtype, int64_t i) {
    return dtype == HIPT_F32 ? ((const float*)p)[i] : (float)((const bf16_t*)p)[i];
}
__global__ __launch_bounds__(256) void small_ab_kernel(const void* __restrict__ x, int xdtype, int N, int S1, int S2x2,
                                                       const void* __restrict__ wab, int wdtype,
                                                       const float* __restrict__ bab, float* __restrict__ ab) {
    const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (i >= (int64_t)N * S2x2) return;
    const int m = (int)(i / S2x2), j = (int)(i % S2x2);
    float a = bab[j];
    for (int k = 0; k < S1; ++k) a += ld_any(x, xdtype, (int64_t)m * S1 + k) * ld_any(wab, wdtype, (int64_t)j * S1 + k);
    ab[i] = a;
}

__global__ __launch_bounds__(256) void max_kernel(const float* __restrict__ A, int N, float* __restrict__ out) {
    __shared__ float red[4];
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, A[i]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// partial (max, sum, acc[S1]) per workgroup over rows [blk*rows_per, ...), h1 fp32 [N,S1]
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ A, const float* __restrict__ h1, int N, int S1,
                                                   const float* __restrict__ gmax, int rows_per,
                                                   float* __restrict__ partials) {
    const int r0 = blockIdx.x * rows_per, r1 = min(N, r0 + rows_per);
    const float mx = gmax[0];
    float* pw = partials + (int64_t)blockIdx.x * (2 + S1);
    __shared__ float red[4];
    float ls = 0.f;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) ls += expf(A[r] - mx);
    ls = wave_sum(ls);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ls;
    __syncthreads();
    if (threadIdx.x == 0) {
        pw[0] = mx;
        pw[1] = red[0] + red[1] + red[2] + red[3];
    }
    for (int c = threadIdx.x; c < S1; c += 256) {
        float a = 0.f;
        for (int r = r0; r < r1; ++r) a += expf(A[r] - mx) * h1[(int64_t)r * S1 + c];
        pw[2 + c] = a;
    }
}

__global__ __launch_bounds__(256) void gather_h1_kernel(const void* bag, int dtype, int S0, int S1, const void* w1,
                                                        const float* b1, const int64_t* idx, float* out) {
    // one workgroup per selected row; thread c computes h1[c]
    const int64_t row = idx[blockIdx.x];
    for (int c = threadIdx.x; c < S1; c += 256) {
        float a = b1[c];
        if (dtype == HIPT_F32) {
            const float* x = (const float*)bag + row * S0;
            const float* w = (const float*)w1 + (int64_t)c * S0;
            for (int k = 0; k < S0; ++k) a += x[k] * w[k];
        } else {
            const bf16_t* x = (const bf16_t*)bag + row * S0;
            const bf16_t* w = (const bf16_t*)w1 + (int64_t)c * S0;
            for (int k = 0; k < S0; ++k) a += (float)x[k] * (float)w[k];
        }
        out[(int64_t)blockIdx.x * S1 + c] = fmaxf(a, 0.f);
    }
}

template <typename T, int S1, int S2>
int launch_fused(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* partials,
                 int* n_partials, hipStream_t st) {
    using G = AG<T, S1, S2>;
    const int ntiles = (N + TM - 1) / TM;
    int grid = ntiles < 512 ? ntiles : 512;
    auto k = abmil_fused_kernel<T, S1, S2>;
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(abmil) failed");
            return HIPT_E_LAUNCH;
        }
        once.done[dev] = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), G::LDS, st, (const T*)bag, N, w->s0, (const T*)w->w1, w->b1,
                       (const T*)w->wab, w->bab, w->wc, w->bc, A_raw, partials, attention_only);
    HIPT_CHECK_LAUNCH();
    *n_partials = grid;
    return HIPT_OK;
}

}  // namespace

bool hipt_clam_fused_supported(const hipt_clam_weights* w) {
    const int kb = w->dtype == HIPT_F32 ? 32 : 64;
    if (w->s0 % kb) return false;
    const int s1 = w->s1, s2 = w->s2;
    if (w->dtype == HIPT_BF16) return (s1 == 128 || s1 == 64) && (s2 == 64 || s2 == 32 || s2 == 16);
    return (s1 == 128 || s1 == 64 || s1 == 32) && (s2 == 64 || s2 == 32 || s2 == 16);
}

int hipt_clam_fused_launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw,
                           float* partials, int* n_partials, hipStream_t st) {
#define FUSED(TT, A, B) \
    if (w->s1 == A && w->s2 == B) return launch_fused<TT, A, B>(w, bag, N, attention_only, A_raw, partials, n_partials, st);
    if (w->dtype == HIPT_BF16) {
        FUSED(bf16_t, 128, 64) FUSED(bf16_t, 128, 32) FUSED(bf16_t, 128, 16)
        FUSED(bf16_t, 64, 64) FUSED(bf16_t, 64, 32) FUSED(bf16_t, 64, 16)
    } else {
        FUSED(float, 128, 64) FUSED(float, 128, 32) FUSED(float, 128, 16)
        FUSED(float, 64, 64) FUSED(float, 64, 32) FUSED(float, 64, 16)
        FUSED(float, 32, 64) FUSED(float, 32, 32) FUSED(float, 32, 16)
    }
#undef FUSED
    hipt_set_error("clam fused: unsupported widths");
    return HIPT_E_UNSUPPORTED;
}

int hipt_clam_combine_launch(const float* partials, int G, const hipt_clam_weights* w, float* M, float* logits,
                             float* Y_prob, int64_t* Y_hat, hipStream_t st) {
    HIPT_CHECK_ARG(G <= CMB_MAXG, "clam combine: %d partials exceed %d", G, CMB_MAXG);
    const size_t lds = (CMB_MAXG + 9 * (size_t)w->s1 + w->n_classes) * sizeof(float);
    hipLaunchKernelGGL(abmil_combine_kernel, dim3(1), dim3(1024), lds, st, partials, G, w->s1, w->wcls, w->bcls,
                       w->n_classes, M, logits, Y_prob, Y_hat);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_gate_launch(const float* ab, int64_t ld, int N, int S2, const float* wc, const float* bc, float* A,
                     hipStream_t st) {
    hipLaunchKernelGGL(gate_kernel, dim3((N + 3) / 4), dim3(256), 0, st, ab, ld, N, S2, wc, bc, A);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_small_ab_launch(const void* x, int xdtype, int N, int S1, int S2x2, const void* wab, int wdtype,
                         const float* bab, float* ab, hipStream_t st) {
    const int64_t n = (int64_t)N * S2x2;
    hipLaunchKernelGGL(small_ab_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xdtype, N, S1, S2x2, wab,
                       wdtype, bab, ab);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_pool_launch(const float* A, const float* h1, int N, int S1, float* gmax, float* partials, int* n_partials,
                     hipStream_t st) {
    hipLaunchKernelGGL(max_kernel, dim3(1), dim3(256), 0, st, A, N, gmax);
    int rows_per = 256;
    while ((N + rows_per - 1) / rows_per > 1024) rows_per *= 2;
    const int G = (N + rows_per - 1) / rows_per;
    hipLaunchKernelGGL(pool_kernel, dim3(G), dim3(256), 0, st, A, h1, N, S1, gmax, rows_per, partials);
    HIPT_CHECK_LAUNCH();
    *n_partials = G;
    return HIPT_OK;
}

int hipt_gather_h1_launch(const hipt_clam_weights* w, const void* bag, const int64_t* idx, int n_idx, float* out,
                          hipStream_t st) {
    hipLaunchKernelGGL(gather_h1_kernel, dim3(n_idx), dim3(256), 0, st, bag, w->dtype, w->s0, w->s1, w->w1, w->b1, idx, out);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}
