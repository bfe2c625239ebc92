// CLAM_SB / CLAM_MB TRAINING step on the GPU: differentiable forward + hand-written backward of
//   models/model_clam.py:147-191 (CLAM_SB.forward), :226-264 (CLAM_MB.forward), :116-145 (inst_eval / inst_eval_out),
// as driven by utils/core_utils.py:300-348 (train_loop_clam) and :373-426 (train_loop: loss.backward() at :423).
//
// The reference trains this module for thousands of epochs on bags of 15-100 rows (docs/README.md:69,154): ~40 tiny
// PyTorch ops per step, forward + backward, all launch-bound.  Here a step is FOUR launches, fp32 throughout (the
// reference's training precision), any size_dict entry (S1, S2 need not be multiples of 16: `hipt_smallest` is [192,8,4]):
//
//   forward   F1 rows kernel   per 16-row tile: x -> h1 = drop(ReLU(x W1^T + b1)) -> t = tanh(h1 Wa^T + ba), s = sigmoid(h1 Wb^T + bb)
//                              -> A[k] = sum_j wc[k][j] drop(t)_j drop(s)_j + bc[k]     (h1, t, s kept for the backward)
//             F2 pool kernel   softmax over the bag per attention branch, M[k] = softmax(A[k]) h1, bag classifier(s),
//                              Y_prob, Y_hat; top-k / bottom-k instance ids per branch ON DEVICE and the gathered h1 rows
//                              (inst_eval: torch.topk + index_select in the reference)
//   backward  B1 rows kernel   per 16-row tile: dA from (dlogits, softmax), gate derivatives, dh1 (pooling + gate GEMM +
//                              instance-branch rows scattered back), dz = dh1 * relu'; per-tile partials of dwc / dbc
//             B2 weights kernel  dW1 = dz^T x, dWa/dWb = d(u|v)^T h1, bias column sums, dwc / dbc / dWcls / dbcls
//
// K = attention branches (1 = CLAM_SB, n_classes = CLAM_MB; "K-branch CLAM_MB in the same kernel").  Dropout: the caller
// passes the scaled masks (0 or 1/(1-p)) it drew with torch's generator, so the RNG stream is the framework's own.
// Everything is deterministic (no float atomics) unless the bag is so long that the weight-gradient reduction is split
// over workgroups (N > 4096 rows), which uses fp32 atomic adds.
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int TR = 16;    // bag rows per workgroup tile
constexpr int KMAX = 8;   // attention branches / classes handled on chip
constexpr int JB = 8;     // output columns a thread accumulates at a time (x 16 column groups = 128 columns per pass)

__device__ __forceinline__ float sigmoid_t(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_t(float x) { return 1.0f - 2.0f / (expf(2.0f * x) + 1.0f); }  // exact limits, ~1e-7 abs
__device__ __forceinline__ float dot4(const f32x4& a, const f32x4& b) {
    return __builtin_fmaf(a[3], b[3], __builtin_fmaf(a[2], b[2], __builtin_fmaf(a[1], b[1], a[0] * b[0])));
}
__host__ __device__ __forceinline__ int pad4(int n) { return n + 4; }  // LDS row stride: 16 rows 4 banks apart (2-way at worst)

// ---- tile loads / stores: [TR][ncol] fp32 between global (row stride ld) and LDS (row stride pad4(ncol)); rows >= nrows are zero
__device__ __forceinline__ void tile_load(const float* __restrict__ g, int64_t ld, int nrows, int ncol, float* lds) {
    const int n4 = ncol >> 2, ldp = pad4(ncol);
    for (int i = threadIdx.x; i < TR * n4; i += 256) {
        const int r = i / n4, c = (i - r * n4) << 2;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows) v = *(const f32x4*)(g + (int64_t)r * ld + c);
        *(f32x4*)(lds + r * ldp + c) = v;
    }
}
__device__ __forceinline__ void tile_store(float* __restrict__ g, int64_t ld, int nrows, int ncol, const float* lds) {
    const int n4 = ncol >> 2, ldp = pad4(ncol);
    for (int i = threadIdx.x; i < TR * n4; i += 256) {
        const int r = i / n4, c = (i - r * n4) << 2;
        if (r < nrows) *(f32x4*)(g + (int64_t)r * ld + c) = *(const f32x4*)(lds + r * ldp + c);
    }
}

// acc[jj] += <xrow[0..Kd), W[c0 + 16 jj][0..Kd)>  for the columns c0 + 16 jj < ncols  ("NT": both operands K-contiguous)
__device__ __forceinline__ void dot_rows(const float* xrow, int Kd, const float* __restrict__ W, int ldw, int c0, int ncols, float (&acc)[JB]) {
    for (int k = 0; k < Kd; k += 4) {
        const f32x4 xv = *(const f32x4*)(xrow + k);
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int c = c0 + 16 * jj;
            if (c < ncols) acc[jj] += dot4(xv, *(const f32x4*)(W + (int64_t)c * ldw + k));
        }
    }
}

struct TrainW {  // device pointers, fp32
    const float *w1, *b1, *wa, *ba, *wb, *bb, *wc, *bc, *wcls, *bcls;
};
struct TrainDims {
    int N, S0, S1, S2, K, C, multi;
};

// =====================================================================================================================
// F1: rows.  LDS: xs [TR][S0+4] | hs [TR][S1+4] | ts [TR][S2+4] | ss [TR][S2+4] | pa [16 cg][KMAX][TR]
// =====================================================================================================================
__global__ __launch_bounds__(256) void clam_train_fwd_rows(const float* __restrict__ bag, TrainDims d, TrainW w, const float* __restrict__ m1,
                                                           const float* __restrict__ ma, const float* __restrict__ mb, float* __restrict__ h1,
                                                           float* __restrict__ tt, float* __restrict__ ss_out, float* __restrict__ A_raw) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S0 = d.S0, S1 = d.S1, S2 = d.S2, K = d.K;
    float* xs = sm;
    float* hs = xs + TR * pad4(S0);
    float* ts = hs + TR * pad4(S1);
    float* ss = ts + TR * pad4(S2);
    float* pa = ss + TR * pad4(S2);
    const int tid = threadIdx.x, r = tid & 15, cg = tid >> 4;
    const int row0 = blockIdx.x * TR, nrows = min(TR, d.N - row0);
    tile_load(bag + (int64_t)row0 * S0, S0, nrows, S0, xs);
    __syncthreads();
    // ---- h1 = drop(ReLU(x W1^T + b1))  (attention_net.0 / .1 / dropout, model_clam.py:83-87) ----
    for (int c0 = cg; c0 < S1; c0 += 16 * JB) {
        float acc[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) acc[jj] = 0.f;
        dot_rows(xs + r * pad4(S0), S0, w.w1, S0, c0, S1, acc);
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int c = c0 + 16 * jj;
            if (c < S1) {
                float h = fmaxf(acc[jj] + w.b1[c], 0.f);
                if (m1 && r < nrows) h *= m1[(int64_t)(row0 + r) * S1 + c];
                hs[r * pad4(S1) + c] = r < nrows ? h : 0.f;
            }
        }
    }
    __syncthreads();
    tile_store(h1 + (int64_t)row0 * S1, S1, nrows, S1, hs);
    // ---- t = tanh(h1 Wa^T + ba), s = sigmoid(h1 Wb^T + bb); gate; A[k] partial sums (Attn_Net_Gated.forward :59-64) ----
    float pk[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) pk[k] = 0.f;
    for (int c0 = cg; c0 < S2; c0 += 16 * JB) {
        float au[JB], av[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) au[jj] = av[jj] = 0.f;
        dot_rows(hs + r * pad4(S1), S1, w.wa, S1, c0, S2, au);
        dot_rows(hs + r * pad4(S1), S1, w.wb, S1, c0, S2, av);
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
            const int j = c0 + 16 * jj;
            if (j < S2) {
                const float t = tanh_t(au[jj] + w.ba[j]), s = sigmoid_t(av[jj] + w.bb[j]);
                ts[r * pad4(S2) + j] = t;
                ss[r * pad4(S2) + j] = s;
                float g = t * s;
                if (r < nrows) {
                    const int64_t o = (int64_t)(row0 + r) * S2 + j;
                    if (ma) g = (t * ma[o]) * (s * mb[o]);
                }
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) pk[k] = __builtin_fmaf(w.wc[k * S2 + j], g, pk[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < K) pa[(cg * KMAX + k) * TR + r] = pk[k];
    __syncthreads();
    tile_store(tt + (int64_t)row0 * S2, S2, nrows, S2, ts);
    tile_store(ss_out + (int64_t)row0 * S2, S2, nrows, S2, ss);
    if (tid < K * TR) {  // (k, r): the 16 column groups in a fixed order -> deterministic logits
        const int k = tid / TR, rr = tid % TR;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) a += pa[(c * KMAX + k) * TR + rr];
        if (rr < nrows) A_raw[(int64_t)k * d.N + row0 + rr] = a + w.bc[k];
    }
}

// block-wide reductions over 256 threads (red: >= 8 floats of LDS)
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// k_sel largest entries of sign * a[0..N), descending, ties -> lowest index first (torch.topk on distinct values; the
// order among equal values is index-ascending, what a stable sort gives).  ids: int64 [k_sel].  One workgroup.
__device__ void topk_block(const float* __restrict__ a, int N, float sign, int k_sel, int64_t* __restrict__ ids, float* red, int* redi) {
    float pv = INFINITY;
    int pi = -1;
    for (int it = 0; it < k_sel; ++it) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < N; i += 256) {
            const float v = sign * a[i];
            const bool eligible = v < pv || (v == pv && i > pi);
            if (eligible && (v > bv || (v == bv && i < bi))) {
                bv = v;
                bi = i;
            }
        }
        // wave argmax (value desc, index asc), then across the 4 waves
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
            red[threadIdx.x >> 6] = bv;
            redi[threadIdx.x >> 6] = bi;
        }
        __syncthreads();
        bv = red[0];
        bi = redi[0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (red[q] > bv || (red[q] == bv && redi[q] < bi)) {
                bv = red[q];
                bi = redi[q];
            }
        if (threadIdx.x == 0) ids[it] = bi;
        pv = bv;
        pi = bi;
    }
}

// =====================================================================================================================
// F2 for long bags: statistics and pooling over many workgroups (grid = row blocks x branches), then F2 proper with prepooled = 1
// =====================================================================================================================
constexpr int POOL_SPLIT_N = 4096, POOL_ROWS = 512;

__global__ void clam_pool_init(float* __restrict__ stats, float* __restrict__ M, int K, int S1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) {
        stats[2 * i] = -INFINITY;
        stats[2 * i + 1] = 0.f;
    }
    if (i < K * S1) M[i] = 0.f;
}

// max of floats through integer atomics: non-negative values order like signed ints, negative ones reversed like unsigned ints
// (the slot starts at -inf)
// -0.0 is canonicalised to +0.0 first (as a negative bit pattern it would go down the unsigned branch and, being the SMALLEST
// negative pattern, win against every real negative maximum only by accident of the comparison direction; as 0x80000000 on the
// signed branch it would never replace -inf) and a NaN to the positive quiet NaN (0x7fc00000: above +inf as a signed int, so it
// wins and stays; a negative NaN pattern -- 0xffc00000, the default NaN -- is ABOVE every negative float as an unsigned int and
// atomicMin would never store it); the branch is taken on the sign bit.
__device__ __forceinline__ void atomic_max_f32(float* p, float v) {
    v += 0.f;
    if (v != v) v = __int_as_float(0x7fc00000);
    if (__float_as_int(v) >= 0) atomicMax((int*)p, __float_as_int(v));
    else atomicMin((unsigned*)p, __float_as_uint(v));
}

__global__ __launch_bounds__(256) void clam_pool_max(const float* __restrict__ A_raw, int N, float* __restrict__ stats) {
    __shared__ float red[8];
    const int k = blockIdx.y, r0 = blockIdx.x * POOL_ROWS;
    const float* a = A_raw + (int64_t)k * N;
    float mx = -INFINITY;
    for (int i = r0 + threadIdx.x; i < min(N, r0 + POOL_ROWS); i += 256) mx = fmaxf(mx, a[i]);
    mx = block_max(mx, red);
    if (threadIdx.x == 0) atomic_max_f32(&stats[2 * k], mx);
}

// sum exp(a - max) and sum exp(a - max) h1 over this block's rows: thread = (row group, column), row groups folded through LDS
__global__ __launch_bounds__(256) void clam_pool_sum(const float* __restrict__ A_raw, const float* __restrict__ h1, int N, int S1,
                                                     float* __restrict__ stats, float* __restrict__ M) {
    __shared__ float part[256];
    __shared__ float red[8];
    const int k = blockIdx.y, r0 = blockIdx.x * POOL_ROWS, r1 = min(N, r0 + POOL_ROWS), tid = threadIdx.x;
    const float* a = A_raw + (int64_t)k * N;
    const float mx = stats[2 * k];
    float se = 0.f;
    for (int i = r0 + tid; i < r1; i += 256) se += expf(a[i] - mx);
    se = block_sum(se, red);
    if (tid == 0) atomicAdd(&stats[2 * k + 1], se);
    int CW = 1;
    while (CW < S1 && CW < 256) CW <<= 1;
    const int RG = 256 / CW, c0 = tid % CW, rg = tid / CW;
    for (int cb = 0; cb < S1; cb += CW) {
        const int c = cb + c0;
        float acc = 0.f;
        if (c < S1)
            for (int i = r0 + rg; i < r1; i += RG) acc = __builtin_fmaf(expf(a[i] - mx), h1[(int64_t)i * S1 + c], acc);
        __syncthreads();
        part[tid] = acc;
        __syncthreads();
        if (rg == 0 && c < S1) {
            float sum = 0.f;
            for (int q = 0; q < RG; ++q) sum += part[q * CW + c0];
            atomicAdd(&M[(int64_t)k * S1 + c], sum);
        }
    }
}

// =====================================================================================================================
// F2: softmax statistics, pooling, classifier(s), top-k ids + gathered rows.  ONE workgroup (K <= 8 branches in turn).
// =====================================================================================================================
// Long bags (N > POOL_SPLIT_N rows: CLAM_MB / CLAM_SB inference through this path, e.g. 100 000 rows) spread the softmax statistics and
// the pooling over workgroups first (clam_pool_init / clam_pool_max / clam_pool_sum below: one workgroup would take ~1 ms per branch); this
// kernel then only normalises M (prepooled).  Their sums are added with fp32 atomics, like the backward's row reductions of long bags:
// training-sized bags (15 - 100 rows) keep the single-workgroup, bit-reproducible order.
__global__ __launch_bounds__(256) void clam_train_pool(TrainDims d, TrainW w, const float* __restrict__ A_raw, const float* __restrict__ h1,
                                                       float* __restrict__ stats, float* __restrict__ M, float* __restrict__ logits,
                                                       float* __restrict__ Y_prob, int64_t* __restrict__ Y_hat, int k_sel,
                                                       int64_t* __restrict__ ids, float* __restrict__ h1_sel, int prepooled) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [256] column partials | [8] red | [8] redi | [KMAX] logits
    float* part = sm;
    float* red = sm + 256;
    int* redi = (int*)(red + 8);
    float* lg = red + 16;
    const int tid = threadIdx.x, N = d.N, S1 = d.S1, K = d.K;
    int CW = 1;
    while (CW < S1 && CW < 256) CW <<= 1;
    const int RG = 256 / CW, c0 = tid % CW, rg = tid / CW;
    for (int k = 0; k < K; ++k) {
        const float* a = A_raw + (int64_t)k * N;
        if (prepooled) {  // stats[2k] = max, stats[2k+1] = sum exp, M[k] = sum exp(a - max) h1: normalise
            const float inv = 1.0f / stats[2 * k + 1];
            for (int c = tid; c < S1; c += 256) M[(int64_t)k * S1 + c] *= inv;
            if (k_sel > 0) {
                topk_block(a, N, 1.0f, k_sel, ids + (int64_t)(2 * k) * k_sel, red, redi);
                topk_block(a, N, -1.0f, k_sel, ids + (int64_t)(2 * k + 1) * k_sel, red, redi);
            }
            continue;
        }
        float mx = -INFINITY;
        for (int i = tid; i < N; i += 256) mx = fmaxf(mx, a[i]);
        mx = block_max(mx, red);
        float se = 0.f;
        for (int i = tid; i < N; i += 256) se += expf(a[i] - mx);
        se = block_sum(se, red);
        if (tid == 0) {
            stats[2 * k] = mx;
            stats[2 * k + 1] = se;
        }
        const float inv = 1.0f / se;
        for (int cb = 0; cb < S1; cb += CW) {  // M[k][c] = sum_i softmax(A[k])_i h1[i][c]   (torch.mm(A, h), :180 / :247)
            const int c = cb + c0;
            float acc = 0.f;
            if (c < S1)
                for (int i = rg; i < N; i += RG) acc = __builtin_fmaf(expf(a[i] - mx) * inv, h1[(int64_t)i * S1 + c], acc);
            __syncthreads();
            part[tid] = acc;
            __syncthreads();
            if (rg == 0 && c < S1) {
                float s = 0.f;
                for (int q = 0; q < RG; ++q) s += part[q * CW + c0];
                M[(int64_t)k * S1 + c] = s;
            }
        }
        if (k_sel > 0) {  // inst_eval's top-k of A (and of -A) for this branch, ids [K][2][k_sel]; softmax is monotone
            topk_block(a, N, 1.0f, k_sel, ids + (int64_t)(2 * k) * k_sel, red, redi);
            topk_block(a, N, -1.0f, k_sel, ids + (int64_t)(2 * k + 1) * k_sel, red, redi);
        }
    }
    __syncthreads();
    __threadfence_block();
    // bag classifier(s): CLAM_SB logits = Wcls M[0] + b (:181); CLAM_MB logits[c] = Wcls[c] . M[c] + b[c] (:248-250)
    for (int c = tid >> 6; c < d.C; c += 4) {
        const float* mrow = M + (int64_t)(d.multi ? c : 0) * S1;
        float acc = 0.f;
        for (int e = tid & 63; e < S1; e += 64) acc = __builtin_fmaf(mrow[e], w.wcls[(int64_t)c * S1 + e], acc);
        acc = wave_sum(acc);
        if ((tid & 63) == 0) lg[c] = acc + w.bcls[c];
    }
    __syncthreads();
    if (tid == 0) {
        float lm = -INFINITY;
        int arg = 0;
        for (int c = 0; c < d.C; ++c)
            if (lg[c] > lm) {
                lm = lg[c];
                arg = c;
            }
        float se = 0.f;
        for (int c = 0; c < d.C; ++c) se += expf(lg[c] - lm);
        for (int c = 0; c < d.C; ++c) {
            logits[c] = lg[c];
            Y_prob[c] = expf(lg[c] - lm) / se;
        }
        Y_hat[0] = arg;
    }
    if (k_sel > 0 && h1_sel) {  // index_select(h, ids) (:120-122, :138)
        const int R = K * 2 * k_sel, n4 = S1 >> 2;
        for (int i = tid; i < R * n4; i += 256) {
            const int e = i / n4, c = (i - e * n4) << 2;
            *(f32x4*)(h1_sel + (int64_t)e * S1 + c) = *(const f32x4*)(h1 + ids[e] * S1 + c);
        }
    }
}

// =====================================================================================================================
// B1: rows.  LDS: hs [TR][S1+4] | ts, ss [TR][S2+4] | duv [TR][2 S2 + 4] | dMs [KMAX][S1] | dAs, ps [KMAX][TR] |
//            pbuf [16][KMAX][TR] | dotM [KMAX] | (dx only) dzs [TR][S1+4]
// =====================================================================================================================
struct BwdIn {
    const float *dlogits, *dM_ext, *dA_ext, *dh1_sel;
    const int64_t* sel_ids;
    int R;
};

__global__ __launch_bounds__(256) void clam_train_bwd_rows(TrainDims d, TrainW w, BwdIn in, const float* __restrict__ A_raw,
                                                           const float* __restrict__ stats, const float* __restrict__ M,
                                                           const float* __restrict__ h1, const float* __restrict__ tt,
                                                           const float* __restrict__ ss_in, const float* __restrict__ m1,
                                                           const float* __restrict__ ma, const float* __restrict__ mb, float* __restrict__ duv_out,
                                                           float* __restrict__ dz_out, float* __restrict__ wcpart, float* __restrict__ dbag) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int S0 = d.S0, S1 = d.S1, S2 = d.S2, K = d.K, N = d.N;
    float* hs = sm;
    float* ts = hs + TR * pad4(S1);
    float* ss = ts + TR * pad4(S2);
    float* duv = ss + TR * pad4(S2);
    float* dMs = duv + TR * pad4(2 * S2);
    float* dAs = dMs + KMAX * S1;
    float* ps = dAs + KMAX * TR;
    float* pbuf = ps + KMAX * TR;
    float* dotM = pbuf + 16 * KMAX * TR;
    float* dzs = dotM + KMAX;
    const int tid = threadIdx.x, r = tid & 15, cg = tid >> 4;
    const int row0 = blockIdx.x * TR, nrows = min(TR, N - row0);
    tile_load(h1 + (int64_t)row0 * S1, S1, nrows, S1, hs);
    tile_load(tt + (int64_t)row0 * S2, S2, nrows, S2, ts);
    tile_load(ss_in + (int64_t)row0 * S2, S2, nrows, S2, ss);
    // dM[k] = d loss / d M[k]: through the bag classifier(s) (+ what arrived on the 'features' output)
    for (int i = tid; i < K * S1; i += 256) {
        const int k = i / S1, c = i - k * S1;
        float v = 0.f;
        if (d.multi) {
            v = in.dlogits[k] * w.wcls[(int64_t)k * S1 + c];
        } else {
            for (int cl = 0; cl < d.C; ++cl) v = __builtin_fmaf(in.dlogits[cl], w.wcls[(int64_t)cl * S1 + c], v);
        }
        if (in.dM_ext) v += in.dM_ext[i];
        dMs[i] = v;
    }
    __syncthreads();
    // dotM[k] = <dM[k], M[k]>; s[k][r] = <dM[k], h1[r]> (16 partial sums per (k, r), fixed order)
    if (tid < 64 * KMAX) {
        const int k = tid >> 6, l = tid & 63;  // one wave per branch (K <= 4 per pass)
        for (int kk = k; kk < K; kk += 4) {
            float a = 0.f;
            for (int c = l; c < S1; c += 64) a = __builtin_fmaf(dMs[kk * S1 + c], M[(int64_t)kk * S1 + c], a);
            a = wave_sum(a);
            if (l == 0) dotM[kk] = a;
        }
    }
    {
        float part[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) part[k] = 0.f;
        for (int c = cg; c < S1; c += 16) {
            const float hv = hs[r * pad4(S1) + c];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) part[k] = __builtin_fmaf(dMs[k * S1 + c], hv, part[k]);
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) pbuf[(cg * KMAX + k) * TR + r] = part[k];
    }
    __syncthreads();
    if (tid < K * TR) {  // dA[k][r] = p (s - <dM, M>) + dA_ext      (softmax over N, :154, then torch.mm :180)
        const int k = tid / TR, rr = tid % TR;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) s += pbuf[(c * KMAX + k) * TR + rr];
        float p = 0.f, dA = 0.f;
        if (rr < nrows) {
            const int64_t o = (int64_t)k * N + row0 + rr;
            p = expf(A_raw[o] - stats[2 * k]) / stats[2 * k + 1];
            dA = p * (s - dotM[k]);
            if (in.dA_ext) dA += in.dA_ext[o];
        }
        ps[k * TR + rr] = p;
        dAs[k * TR + rr] = dA;
    }
    __syncthreads();
    // gate backward: du = dgate * drop(s) * ma * (1 - t^2), dv = dgate * drop(t) * mb * s (1 - s); dwc / dbc partials of this tile
    float* wcp = wcpart + (int64_t)blockIdx.x * (K * S2 + K);
    for (int c0 = cg; c0 < S2; c0 += 16) {
        const int j = c0;
        const float t = ts[r * pad4(S2) + j], s = ss[r * pad4(S2) + j];
        float fa = 1.f, fb = 1.f;
        if (ma && r < nrows) {
            const int64_t o = (int64_t)(row0 + r) * S2 + j;
            fa = ma[o];
            fb = mb[o];
        }
        const float ad = t * fa, bd = s * fb, g = ad * bd;
        float dg = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const float dA = dAs[k * TR + r];
                dg = __builtin_fmaf(dA, w.wc[k * S2 + j], dg);
                float v = dA * g;  // sum over the 16 rows of the tile (the 16 lanes that share this column group)
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (r == 0) wcp[k * S2 + j] = v;
            }
        duv[r * pad4(2 * S2) + j] = r < nrows ? dg * bd * fa * (1.0f - t * t) : 0.f;
        duv[r * pad4(2 * S2) + S2 + j] = r < nrows ? dg * ad * fb * s * (1.0f - s) : 0.f;
    }
    if (tid < K) {
        float v = 0.f;
        for (int rr = 0; rr < TR; ++rr) v += dAs[tid * TR + rr];
        wcp[K * S2 + tid] = v;
    }
    __syncthreads();
    tile_store(duv_out + (int64_t)row0 * 2 * S2, 2 * S2, nrows, 2 * S2, duv);
    // dh1 = sum_k p_k dM[k] + du Wa + dv Wb (+ rows selected by the instance branch); dz = dh1 * m1 * [h1 > 0]
    {
        const int rr = tid >> 4, cq = tid & 15;  // row, float4 column group: 16 lanes cover 64 contiguous columns
        for (int cb = 4 * cq; cb < S1; cb += 64) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) acc += ps[k * TR + rr] * *(const f32x4*)(dMs + k * S1 + cb);
            for (int j = 0; j < S2; ++j) {
                acc += duv[rr * pad4(2 * S2) + j] * *(const f32x4*)(w.wa + (int64_t)j * S1 + cb);
                acc += duv[rr * pad4(2 * S2) + S2 + j] * *(const f32x4*)(w.wb + (int64_t)j * S1 + cb);
            }
            for (int e = 0; e < in.R; ++e)
                if (in.sel_ids[e] == row0 + rr) acc += *(const f32x4*)(in.dh1_sel + (int64_t)e * S1 + cb);
            f32x4 hv = *(const f32x4*)(hs + rr * pad4(S1) + cb), mk = {1.f, 1.f, 1.f, 1.f};
            if (m1 && rr < nrows) mk = *(const f32x4*)(m1 + (int64_t)(row0 + rr) * S1 + cb);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = (hv[e] > 0.f && rr < nrows) ? acc[e] * mk[e] : 0.f;
            if (rr < nrows) *(f32x4*)(dz_out + (int64_t)(row0 + rr) * S1 + cb) = acc;
            if (dbag) *(f32x4*)(dzs + rr * pad4(S1) + cb) = acc;
        }
        if (dbag) {  // d bag = dz W1   (only when the bag itself requires a gradient)
            __syncthreads();
            for (int cb = 4 * cq; cb < S0; cb += 64) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int c = 0; c < S1; ++c) acc += dzs[rr * pad4(S1) + c] * *(const f32x4*)(w.w1 + (int64_t)c * S0 + cb);
                if (rr < nrows) *(f32x4*)(dbag + (int64_t)(row0 + rr) * S0 + cb) = acc;
            }
        }
    }
}

// =====================================================================================================================
// B2: weight gradients.  Jobs 0..2 are "TN" products out[m][n] = sum_i P[i][m] Q[i][n] over the bag rows in 16 x 64
// output tiles (blockIdx.y splits the rows: atomics when gridDim.y > 1); the last workgroup of the grid does the small
// ones (bias column sums would cost a pass of their own: they ride as an extra Q column of ones... kept simple: here).
// =====================================================================================================================
struct TnJob {
    const float *P, *Q;
    float* out;
    int ldp, ldq, ldo, Mo, No, tiles_n, tile0;
};
struct WJobs {
    TnJob j[3];
    int ntiles;
};
struct SmallOut {
    float *db1, *dba, *dbb, *dwc, *dbc, *dwcls, *dbcls;
};

__global__ __launch_bounds__(256) void clam_train_bwd_weights(TrainDims d, WJobs jobs, const float* __restrict__ dz, const float* __restrict__ duv,
                                                              const float* __restrict__ wcpart, int G, const float* __restrict__ dlogits,
                                                              const float* __restrict__ M, SmallOut so) {
    __shared__ __attribute__((aligned(16))) float Ps[16 * 16];
    __shared__ __attribute__((aligned(16))) float Qs[16 * 64];
    const int tid = threadIdx.x, N = d.N;
    if ((int)blockIdx.x < jobs.ntiles) {
        int ji = 0;
        if ((int)blockIdx.x >= jobs.j[1].tile0) ji = 1;
        if ((int)blockIdx.x >= jobs.j[2].tile0) ji = 2;
        const TnJob J = jobs.j[ji];
        const int t = blockIdx.x - J.tile0, m0 = (t / J.tiles_n) * 16, n0 = (t % J.tiles_n) * 64;
        const int per = (N + gridDim.y - 1) / gridDim.y;
        const int i_beg = blockIdx.y * per, i_end = min(N, i_beg + per);
        const int m = tid & 15, n4 = tid >> 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int i0 = i_beg; i0 < i_end; i0 += 16) {
            {
                const int ii = tid >> 4, mm = tid & 15;
                float v = 0.f;
                if (i0 + ii < i_end && m0 + mm < J.Mo) v = J.P[(int64_t)(i0 + ii) * J.ldp + m0 + mm];
                Ps[ii * 16 + mm] = v;
                f32x4 q = {0.f, 0.f, 0.f, 0.f};
                const int nn = n0 + 4 * mm;
                if (i0 + ii < i_end && nn < J.No) q = *(const f32x4*)(J.Q + (int64_t)(i0 + ii) * J.ldq + nn);  // (No % 4 == 0)
                *(f32x4*)(Qs + ii * 64 + 4 * mm) = q;
            }
            __syncthreads();
#pragma unroll
            for (int ii = 0; ii < 16; ++ii) acc += Ps[ii * 16 + m] * *(const f32x4*)(Qs + ii * 64 + 4 * n4);
            __syncthreads();
        }
        const int mo = m0 + m, no = n0 + 4 * n4;
        if (mo < J.Mo && no < J.No) {
            float* o = J.out + (int64_t)mo * J.ldo + no;
            if (gridDim.y == 1) {
                *(f32x4*)o = acc;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(o + e, acc[e]);
            }
        }
        return;
    }
    if (blockIdx.y != 0) return;
    // ---- the small gradients: one workgroup ----
    const int S1 = d.S1, S2 = d.S2, K = d.K;
    for (int c = tid; c < S1; c += 256) {  // db1 = column sums of dz
        float a = 0.f;
        for (int i = 0; i < N; ++i) a += dz[(int64_t)i * S1 + c];
        so.db1[c] = a;
    }
    for (int j = tid; j < 2 * S2; j += 256) {  // dba | dbb = column sums of (du | dv)
        float a = 0.f;
        for (int i = 0; i < N; ++i) a += duv[(int64_t)i * 2 * S2 + j];
        (j < S2 ? so.dba[j] : so.dbb[j - S2]) = a;
    }
    for (int e = tid; e < K * S2 + K; e += 256) {  // dwc [K][S2] | dbc [K]: the tiles' partials in tile order
        float a = 0.f;
        for (int g = 0; g < G; ++g) a += wcpart[(int64_t)g * (K * S2 + K) + e];
        (e < K * S2 ? so.dwc[e] : so.dbc[e - K * S2]) = a;
    }
    for (int e = tid; e < d.C * S1; e += 256) {  // dWcls[c] = dlogits[c] * M[c or 0]
        const int c = e / S1, x = e - c * S1;
        so.dwcls[e] = dlogits[c] * M[(int64_t)(d.multi ? c : 0) * S1 + x];
    }
    for (int c = tid; c < d.C; c += 256) so.dbcls[c] = dlogits[c];
}

__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ A, int N, int k_sel, int64_t* __restrict__ ids) {
    __shared__ float red[8];
    __shared__ int redi[8];
    const float* a = A + (int64_t)blockIdx.x * N;
    topk_block(a, N, 1.0f, k_sel, ids + (int64_t)(2 * blockIdx.x) * k_sel, red, redi);
    topk_block(a, N, -1.0f, k_sel, ids + (int64_t)(2 * blockIdx.x + 1) * k_sel, red, redi);
}

TrainDims dims_of(const hipt_clam_train_weights* w, int N) { return TrainDims{N, w->s0, w->s1, w->s2, w->n_att, w->n_classes, w->multi_branch}; }
TrainW ptrs_of(const hipt_clam_train_weights* w) { return TrainW{w->w1, w->b1, w->wa, w->ba, w->wb, w->bb, w->wc, w->bc, w->wcls, w->bcls}; }

int check_train(const hipt_clam_train_weights* w, int N) {
    HIPT_CHECK_ARG(w != nullptr && N > 0, "clam_train: null weights / empty bag");
    HIPT_CHECK_ARG(w->s0 > 0 && w->s1 > 0 && w->s2 > 0 && w->s0 % 4 == 0 && w->s1 % 4 == 0 && w->s2 % 4 == 0,
                   "clam_train: widths [%d,%d,%d] must be positive multiples of 4", w->s0, w->s1, w->s2);
    HIPT_CHECK_ARG(w->n_att >= 1 && w->n_att <= KMAX && w->n_classes >= 1 && w->n_classes <= KMAX, "clam_train: %d branches / %d classes (at most %d)", w->n_att,
                   w->n_classes, KMAX);
    HIPT_CHECK_ARG(!w->multi_branch || w->n_att == w->n_classes, "clam_train: CLAM_MB has one attention branch per class");
    HIPT_CHECK_ARG(w->w1 && w->b1 && w->wa && w->ba && w->wb && w->bb && w->wc && w->bc && w->wcls && w->bcls, "clam_train: null weight pointer");
    return HIPT_OK;
}

size_t fwd_lds(const hipt_clam_train_weights* w) {
    return (size_t)(TR * (pad4(w->s0) + pad4(w->s1) + 2 * pad4(w->s2)) + 16 * KMAX * TR) * sizeof(float);
}
size_t bwd_lds(const hipt_clam_train_weights* w, bool dbag) {
    return (size_t)(TR * (pad4(w->s1) + 2 * pad4(w->s2) + pad4(2 * w->s2)) + KMAX * w->s1 + 2 * KMAX * TR + 16 * KMAX * TR + KMAX +
                    (dbag ? TR * pad4(w->s1) : 0)) *
           sizeof(float);
}

}  // namespace

extern "C" {

int hipt_clam_train_shape_supported(int s0, int s1, int s2, int n_att, int n_classes, int need_dbag) {
    if (s0 <= 0 || s1 <= 0 || s2 <= 0 || (s0 | s1 | s2) % 4 != 0 || n_att < 1 || n_att > KMAX || n_classes < 1 || n_classes > KMAX) return 0;
    hipt_clam_train_weights w;
    memset(&w, 0, sizeof(w));
    w.s0 = s0; w.s1 = s1; w.s2 = s2; w.n_att = n_att; w.n_classes = n_classes;
    return fwd_lds(&w) <= 160 * 1024 && bwd_lds(&w, need_dbag != 0) <= 160 * 1024;
}

size_t hipt_clam_train_workspace_bytes(const hipt_clam_train_weights* w, int N) {
    if (!w || N <= 0) return 0;
    const size_t G = (size_t)(N + TR - 1) / TR;
    // duv [N, 2 S2] | dz [N, S1] | per-tile (dwc, dbc) partials
    return (((size_t)N * 2 * w->s2 + (size_t)N * w->s1 + G * ((size_t)w->n_att * w->s2 + w->n_att)) * sizeof(float) + 1023) & ~(size_t)255;
}

int hipt_clam_train_forward(const hipt_clam_train_weights* w, const float* bag, int N, const float* m1, const float* ma, const float* mb, float* h1,
                            float* t, float* s, float* A_raw, float* stats, float* M, float* logits, float* Y_prob, int64_t* Y_hat, int k_sample,
                            int64_t* topk_ids, float* h1_sel, void* stream) {
    int rc = check_train(w, N);
    if (rc) return rc;
    HIPT_CHECK_ARG(bag && h1 && t && s && A_raw && stats && M && logits && Y_prob && Y_hat, "clam_train_forward: null buffer");
    HIPT_CHECK_ARG((ma == nullptr) == (mb == nullptr), "clam_train_forward: the two gate dropout masks come together");
    HIPT_CHECK_ARG(k_sample >= 0 && (k_sample == 0 || topk_ids), "clam_train_forward: k_sample without an id buffer");
    if (k_sample > N) {  // torch.topk raises "selected index k out of range" (model_clam.py:120)
        hipt_set_error("clam_train_forward: k_sample=%d exceeds the bag's %d rows", k_sample, N);
        return HIPT_E_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const TrainDims d = dims_of(w, N);
    const TrainW p = ptrs_of(w);
    const size_t lds = fwd_lds(w);
    if (lds > 160 * 1024) {
        hipt_set_error("clam_train_forward: widths [%d,%d,%d] need %zu B of LDS", w->s0, w->s1, w->s2, lds);
        return HIPT_E_UNSUPPORTED;
    }
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)clam_train_fwd_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)clam_train_bwd_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(clam_train) failed");
            return HIPT_E_LAUNCH;
        }
        once.done[dev] = true;
    }
    hipLaunchKernelGGL(clam_train_fwd_rows, dim3((N + TR - 1) / TR), dim3(256), lds, st, bag, d, p, m1, ma, mb, h1, t, s, A_raw);
    HIPT_CHECK_LAUNCH();
    const int prepooled = N > POOL_SPLIT_N ? 1 : 0;
    if (prepooled) {
        const int K = w->n_att, G = (N + POOL_ROWS - 1) / POOL_ROWS;
        hipLaunchKernelGGL(clam_pool_init, dim3((K * w->s1 + 255) / 256), dim3(256), 0, st, stats, M, K, w->s1);
        hipLaunchKernelGGL(clam_pool_max, dim3(G, K), dim3(256), 0, st, A_raw, N, stats);
        hipLaunchKernelGGL(clam_pool_sum, dim3(G, K), dim3(256), 0, st, A_raw, h1, N, w->s1, stats, M);
        HIPT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(clam_train_pool, dim3(1), dim3(256), (256 + 16 + KMAX) * sizeof(float), st, d, p, A_raw, h1, stats, M, logits, Y_prob, Y_hat, k_sample,
                       topk_ids, h1_sel, prepooled);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_clam_train_backward(const hipt_clam_train_weights* w, const float* bag, int N, const float* m1, const float* ma, const float* mb,
                             const float* h1, const float* t, const float* s, const float* A_raw, const float* stats, const float* M,
                             const float* dlogits, const float* dA_raw, const float* dM, const int64_t* sel_ids, const float* dh1_sel, int n_sel,
                             const hipt_clam_train_grads* g, void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_train(w, N);
    if (rc) return rc;
    HIPT_CHECK_ARG(bag && h1 && t && s && A_raw && stats && M && dlogits && g, "clam_train_backward: null buffer");
    HIPT_CHECK_ARG(g->dw1 && g->db1 && g->dwa && g->dba && g->dwb && g->dbb && g->dwc && g->dbc && g->dwcls && g->dbcls, "clam_train_backward: null gradient buffer");
    HIPT_CHECK_ARG(n_sel == 0 || (sel_ids && dh1_sel), "clam_train_backward: instance rows without ids / gradients");
    if (ws_bytes < hipt_clam_train_workspace_bytes(w, N) || ((uintptr_t)workspace & 255)) {
        hipt_set_error("clam_train_backward: workspace %zu B too small / unaligned (need %zu)", ws_bytes, hipt_clam_train_workspace_bytes(w, N));
        return HIPT_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const TrainDims d = dims_of(w, N);
    const TrainW p = ptrs_of(w);
    const int G = (N + TR - 1) / TR, S0 = w->s0, S1 = w->s1, S2 = w->s2, K = w->n_att;
    float* duv = (float*)workspace;
    float* dz = duv + (size_t)N * 2 * S2;
    float* wcpart = dz + (size_t)N * S1;
    const size_t lds = bwd_lds(w, g->dbag != nullptr);
    if (lds > 160 * 1024) {
        hipt_set_error("clam_train_backward: widths [%d,%d,%d] need %zu B of LDS", S0, S1, S2, lds);
        return HIPT_E_UNSUPPORTED;
    }
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)clam_train_bwd_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(clam_train_bwd_rows) failed");
            return HIPT_E_LAUNCH;
        }
        once.done[dev] = true;
    }
    BwdIn in{dlogits, dM, dA_raw, dh1_sel, sel_ids, n_sel};
    hipLaunchKernelGGL(clam_train_bwd_rows, dim3(G), dim3(256), lds, st, d, p, in, A_raw, stats, M, h1, t, s, m1, ma, mb, duv, dz, wcpart, g->dbag);
    HIPT_CHECK_LAUNCH();
    WJobs jobs;
    auto mk = [](const float* P, int ldp, int Mo, const float* Q, int ldq, int No, float* out, int tile0) {
        TnJob j;
        j.P = P; j.Q = Q; j.out = out; j.ldp = ldp; j.ldq = ldq; j.ldo = No; j.Mo = Mo; j.No = No;
        j.tiles_n = (No + 63) / 64;
        j.tile0 = tile0;
        return j;
    };
    jobs.j[0] = mk(dz, S1, S1, bag, S0, S0, g->dw1, 0);                                      // dW1 = dz^T x
    int t0 = ((S1 + 15) / 16) * jobs.j[0].tiles_n;
    jobs.j[1] = mk(duv, 2 * S2, S2, h1, S1, S1, g->dwa, t0);                                 // dWa = du^T h1
    t0 += ((S2 + 15) / 16) * jobs.j[1].tiles_n;
    jobs.j[2] = mk(duv + S2, 2 * S2, S2, h1, S1, S1, g->dwb, t0);                            // dWb = dv^T h1
    t0 += ((S2 + 15) / 16) * jobs.j[2].tiles_n;
    jobs.ntiles = t0;
    int nsplit = 1;
    if (N > 4096) {  // long bags: the row reduction is split over workgroups, partial tiles added with fp32 atomics
        nsplit = (N + 4095) / 4096;
        if (nsplit > 64) nsplit = 64;
        if (hipMemsetAsync(g->dw1, 0, (size_t)S1 * S0 * 4, st) != hipSuccess || hipMemsetAsync(g->dwa, 0, (size_t)S2 * S1 * 4, st) != hipSuccess ||
            hipMemsetAsync(g->dwb, 0, (size_t)S2 * S1 * 4, st) != hipSuccess) {
            hipt_set_error("clam_train_backward: hipMemsetAsync failed");
            return HIPT_E_LAUNCH;
        }
    }
    SmallOut so{g->db1, g->dba, g->dbb, g->dwc, g->dbc, g->dwcls, g->dbcls};
    hipLaunchKernelGGL(clam_train_bwd_weights, dim3(jobs.ntiles + 1, nsplit), dim3(256), 0, st, d, jobs, dz, duv, wcpart, G, dlogits, M, so);
    HIPT_CHECK_LAUNCH();
    (void)K;
    return HIPT_OK;
}

int hipt_topk_rows(const float* A, int rows, int N, int k, int64_t* ids, void* stream) {
    HIPT_CHECK_ARG(A && ids && rows > 0 && N > 0 && k > 0, "topk_rows: null/empty argument");
    if (k > N) {
        hipt_set_error("topk_rows: k=%d exceeds the row length %d", k, N);
        return HIPT_E_BADARG;
    }
    hipLaunchKernelGGL(topk_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, A, N, k, ids);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

}  // extern "C"
