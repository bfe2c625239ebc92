// LayerNorm (vision_transformer.py:138,142,195), [CLS] row initialisation (:240-241,244) and the
// fp32 -> bf16 conversion used by the bf16 mode.  All HBM-bound: one wave per row, coalesced reads,
// wavefront (64-lane) shuffle reductions, fp32 statistics with the two-pass (mean, then centred
// variance) form torch uses.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int LN_MAX_E = 32;  // D <= 2048

template <typename TO, int V> __device__ __forceinline__ void ln_store(TO* p, const float* v);
template <> __device__ __forceinline__ void ln_store<float, 1>(float* p, const float* v) { p[0] = v[0]; }
template <> __device__ __forceinline__ void ln_store<float, 2>(float* p, const float* v) { *(f32x2*)p = f32x2{v[0], v[1]}; }
template <> __device__ __forceinline__ void ln_store<bf16_t, 1>(bf16_t* p, const float* v) { p[0] = (bf16_t)v[0]; }
template <> __device__ __forceinline__ void ln_store<bf16_t, 2>(bf16_t* p, const float* v) {
    *(uint32_t*)p = pack_bf16x2(v[0], v[1]);
}

// one wave per row; lane l owns elements V*l + 64*V*i (+0..V-1): every load/store instruction of the
// wave is one contiguous 256*V-byte run
template <typename TO, int V>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, int64_t x_stride,
                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                 TO* __restrict__ out, int64_t out_stride, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * x_stride;
    const int E = D / (64 * V);
    float v[LN_MAX_E / V][V];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_E / V; ++i)
        if (i < E) {
            if constexpr (V == 2) {
                const f32x2 t = *(const f32x2*)(xr + 2 * lane + 128 * i);
                v[i][0] = t[0];
                v[i][1] = t[1];
            } else {
                v[i][0] = xr[lane + 64 * i];
            }
#pragma unroll
            for (int e = 0; e < V; ++e) s += v[i][e];
        }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_E / V; ++i)
        if (i < E) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float c = v[i][e] - mean;
                q += c * c;
            }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    TO* orow = out + (int64_t)row * out_stride;
#pragma unroll
    for (int i = 0; i < LN_MAX_E / V; ++i)
        if (i < E) {
            const int d = V * lane + 64 * V * i;
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = (v[i][e] - mean) * rstd * w[d + e] + b[d + e];
            ln_store<TO, V>(orow + d, o);
        }
}

__global__ void cls_init_kernel(float* x, const float* cls, const float* pos, int nseq, int ntok, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseq * D) return;
    const int s = i / D, d = i % D;
    x[(int64_t)s * ntok * D + d] = cls[d] + pos[d];
}

// one wave per sequence: the [CLS] row (cls + pos[0]: the same 384 values for every sequence) into the fp32 activation image, its LayerNorm into
// the bf16 one (kernels.h: fragment F = rows [16 F, 16 F + 16), row li, 16-byte chunk g + 4 c at c * 512 + (16 g + li) * 8 bf16 elements)
__global__ __launch_bounds__(64) void cls_init_img_kernel(float* __restrict__ x, bf16_t* __restrict__ xn, const float* __restrict__ cls, const float* __restrict__ pos,
                                                          const float* __restrict__ gw, const float* __restrict__ gb, float eps, int nseq, int ntok) {
#pragma clang fp contract(off)
    constexpr int D = 384;
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= nseq) return;
    float v[6], sum = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        v[j] = cls[lane + 64 * j] + pos[lane + 64 * j];
        sum += v[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float mean = sum / (float)D;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) var = __builtin_fmaf(v[j] - mean, v[j] - mean, var);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o, 64);
    const float rstd = 1.0f / sqrtf(var / (float)D + eps);
    const int64_t R = (int64_t)s * ntok, F = R >> 4;
    const int li = (int)(R & 15);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int d = lane + 64 * j, ch = d >> 3;
        x[F * 6144 + (ch >> 2) * 512 + ((d >> 2) & 1) * 256 + (16 * (ch & 3) + li) * 4 + (d & 3)] = v[j];
        xn[F * 6144 + (ch >> 2) * 512 + (16 * (ch & 3) + li) * 8 + (d & 7)] = (bf16_t)__builtin_fmaf((v[j] - mean) * rstd, gw[d], gb[d]);
    }
}

__global__ void f32_to_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = *(const f32x4*)(in + i * 8), c = *(const f32x4*)(in + i * 8 + 4);
        u32x4 o;
        o[0] = pack_bf16x2(a[0], a[1]);
        o[1] = pack_bf16x2(a[2], a[3]);
        o[2] = pack_bf16x2(c[0], c[1]);
        o[3] = pack_bf16x2(c[2], c[3]);
        *(u32x4*)(out + i * 8) = o;
    }
}

// ToTensor + Normalize(mean 0.5, std 0.5) (hipt_model_utils.py:113-118): (x / 255 - 0.5) / 0.5 in fp32, exactly the
// torchvision arithmetic (true division, then subtract, then divide).  16 pixels of one channel per thread.
// HWC = 0: src [n, 3, plane] ; HWC = 1: src [n, plane, 3] (interleaved RGB) -> dst [n, 3, plane].
template <typename TO> __device__ __forceinline__ void store16(TO* dst, const float (&v)[16]);
template <> __device__ __forceinline__ void store16<float>(float* dst, const float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
}
template <> __device__ __forceinline__ void store16<bf16_t>(bf16_t* dst, const float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(v[8 * q + 2 * e], v[8 * q + 2 * e + 1]);
        *(u32x4*)(dst + 8 * q) = o;
    }
}
__device__ __forceinline__ float u8_norm(uint32_t b) { return ((float)b / 255.0f - 0.5f) / 0.5f; }

template <typename TO, int HWC>
__global__ void u8_norm_kernel(const uint8_t* __restrict__ src, TO* __restrict__ dst, int64_t nimg, int64_t plane) {
    const int64_t p16 = plane / 16;  // groups of 16 pixels per plane
    const int64_t total = nimg * p16 * (HWC ? 1 : 3);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if constexpr (HWC == 0) {
            const u32x4 raw = *(const u32x4*)(src + i * 16);
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = u8_norm((raw[e >> 2] >> (8 * (e & 3))) & 0xffu);
            store16<TO>(dst + i * 16, v);
        } else {
            const int64_t img = i / p16, g16 = i % p16;
            const uint8_t* sp = src + (img * plane + g16 * 16) * 3;
            const u32x4 r0 = *(const u32x4*)sp, r1 = *(const u32x4*)(sp + 16), r2 = *(const u32x4*)(sp + 32);
            const uint32_t w[12] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3], r2[0], r2[1], r2[2], r2[3]};
            float v[3][16];
#pragma unroll
            for (int px = 0; px < 16; ++px)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int byte = 3 * px + c;
                    v[c][px] = u8_norm((w[byte >> 2] >> (8 * (byte & 3))) & 0xffu);
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) store16<TO>(dst + (img * 3 + c) * plane + g16 * 16, v[c]);
        }
    }
}

__global__ void add_bf16_kernel(float* __restrict__ out, const float* __restrict__ src, const bf16_t* __restrict__ y, int64_t n8) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 a = *(const f32x4*)(src + i * 8), c = *(const f32x4*)(src + i * 8 + 4);
        if (y) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, *(const u32x4*)(y + i * 8));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[e] += (float)v[e];
                c[e] += (float)v[4 + e];
            }
        }
        *(f32x4*)(out + i * 8) = a;
        *(f32x4*)(out + i * 8 + 4) = c;
    }
}

// Attention of ONE query (token 0, the [CLS] token) per (sequence, head), bf16 qkv, head dim 64: what the last
// ViT-256 block needs when only x[:, 0] is consumed afterwards (vision_transformer.py:253; SURVEY.md 8d allows the
// pruning).  One wave per (b, h): lane l scores keys l, l + 64, ... (fp32 dot products of the bf16 values, exact
// products), wave softmax, then lane d accumulates output dimension d over all keys.
__global__ __launch_bounds__(256) void attn_cls_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, float* __restrict__ probs,
                                                       int nbh, int ntok, int heads, float scale) {
    constexpr int DH = 64, MAXK = 5;  // up to 320 keys
    const int lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= nbh) return;
    const int b = bh / heads, h = bh % heads, D = heads * DH;
    const int64_t tokstride = 3 * (int64_t)D;
    const bf16_t* qrow = qkv + (int64_t)b * ntok * tokstride + h * DH;  // token 0
    const bf16_t* kbase = qrow + D;
    const bf16_t* vbase = qrow + 2 * D;
    float q[DH];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bf16x8 v = __builtin_bit_cast(bf16x8, *(const u32x4*)(qrow + c * 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) q[c * 8 + e] = (float)v[e];
    }
    float sc[MAXK];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        const int key = j * 64 + lane;
        sc[j] = -INFINITY;
        if (key < ntok) {
            const bf16_t* kr = kbase + key * tokstride;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bf16x8 v = __builtin_bit_cast(bf16x8, *(const u32x4*)(kr + c * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) a = __builtin_fmaf(q[c * 8 + e], (float)v[e], a);
            }
            sc[j] = a * scale;
        }
        m = fmaxf(m, sc[j]);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        sc[j] = expf(sc[j] - m);  // exp(-inf) = 0 for the keys past the end
        l += sc[j];
    }
    l = wave_sum(l);
    if (probs) {  // attention of the [CLS] query over all keys (heat-maps read [:, :, 0, 1:] of the last block's map)
        const float inv = 1.0f / l;
#pragma unroll
        for (int j = 0; j < MAXK; ++j)
            if (j * 64 + lane < ntok) probs[(int64_t)bh * ntok + j * 64 + lane] = sc[j] * inv;
        if (!out) return;
    }
    // O = sum_key p[key] * V[key]: lane (kg = lane >> 3, dc = lane & 7) takes keys kg, kg + 8, ... and the 8 dimensions
    // 8 dc .. (one 16-byte load per key: 8 keys x 128 B per wave instruction, all loads independent), then the 8 key
    // groups are summed with three xor-shuffles
    const int kg = lane >> 3, dc = lane & 7;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const int key = j * 64 + ii * 8 + kg;
            const float pk = __shfl(sc[j], ii * 8 + kg, 64);  // 0 for keys past the end
            const int kc = key < ntok ? key : ntok - 1;
            const bf16x8 v = __builtin_bit_cast(bf16x8, *(const u32x4*)(vbase + (int64_t)kc * tokstride + dc * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(pk, (float)v[e], o[e]);
        }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] += __shfl_xor(o[e], 8, 64);
        o[e] += __shfl_xor(o[e], 16, 64);
        o[e] += __shfl_xor(o[e], 32, 64);
    }
    if (kg == 0) {
        const float inv = 1.0f / l;
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(o[2 * e] * inv, o[2 * e + 1] * inv);
        *(u32x4*)(out + (int64_t)b * D + h * DH + dc * 8) = w;
    }
}

// Attention PROBABILITIES of the [CLS] query only, any compute dtype, head dim 32 or 64 (what
// get_last_selfattention(x)[:, :, 0, :] holds, HIPT_4K/hipt_4k.py:143-158, without the [B, heads, N, N] tensor): one
// wave per (sequence, head), lane l scores keys l, l + 64, ... in fp32 (k-ordered fma chain), wave softmax.
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_cls_probs_kernel(const T* __restrict__ qkv, float* __restrict__ probs, int nbh, int ntok, int heads,
                                                             float scale) {
    constexpr int MAXK = 5, EPC = Tr<T>::EPC;  // up to 320 keys; elements per 16-byte chunk
    const int lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= nbh) return;
    const int b = bh / heads, h = bh % heads, D = heads * DH;
    const int64_t tokstride = 3 * (int64_t)D;
    const T* qrow = qkv + (int64_t)b * ntok * tokstride + h * DH;  // token 0
    const T* kbase = qrow + D;
    float q[DH];
#pragma unroll
    for (int c = 0; c < DH / EPC; ++c) {
        const u32x4 raw = *(const u32x4*)(qrow + c * EPC);
        if constexpr (sizeof(T) == 2) {
            const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[c * 8 + e] = (float)v[e];
        } else {
            const f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
            for (int e = 0; e < 4; ++e) q[c * 4 + e] = v[e];
        }
    }
    float sc[MAXK];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        const int key = j * 64 + lane;
        sc[j] = -INFINITY;
        if (key < ntok) {
            const T* kr = kbase + key * tokstride;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < DH / EPC; ++c) {
                const u32x4 raw = *(const u32x4*)(kr + c * EPC);
                if constexpr (sizeof(T) == 2) {
                    const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
                    for (int e = 0; e < 8; ++e) a = __builtin_fmaf(q[c * 8 + e], (float)v[e], a);
                } else {
                    const f32x4 v = __builtin_bit_cast(f32x4, raw);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a = __builtin_fmaf(q[c * 4 + e], v[e], a);
                }
            }
            sc[j] = a * scale;
        }
        m = fmaxf(m, sc[j]);
    }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        sc[j] = expf(sc[j] - m);  // exp(-inf) = 0 for the keys past the end
        l += sc[j];
    }
    l = wave_sum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int j = 0; j < MAXK; ++j)
        if (j * 64 + lane < ntok) probs[(int64_t)bh * ntok + j * 64 + lane] = sc[j] * inv;
}

// dst[s, :] = src[s * ntok, :]  (fp32 rows of D floats: the [CLS] rows of the residual stream)
// img: src is an fp32 activation image (kernels.h; D = 384, seq_stride a multiple of D): row r = s * seq_stride / D sits in
// fragment r / 16 as li = r % 16; its 4 floats at column 4c are chunk c / 2 = g + 4 cc, half c & 1
__global__ void gather_cls_kernel(const float* __restrict__ src, float* __restrict__ dst, int nseq, int64_t seq_stride, int D, int img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one float4
    const int per = D / 4;
    if (i >= nseq * per) return;
    const int s = i / per, c = i % per;
    if (img) {
        const int64_t r = (int64_t)s * seq_stride / D;
        const int ch = c >> 1, h = c & 1, g = ch & 3, cc = ch >> 2, li = (int)(r & 15);
        *(f32x4*)(dst + (int64_t)s * D + 4 * c) = *(const f32x4*)(src + (r >> 4) * (16 * D) + cc * 512 + h * 256 + (g * 16 + li) * 4);
    } else {
        *(f32x4*)(dst + (int64_t)s * D + 4 * c) = *(const f32x4*)(src + (int64_t)s * seq_stride + 4 * c);
    }
}

// the same for bf16 rows of 384 (the LayerNorm'd operands): dst[s, :] = src[s * ntok, :]; img: src is a bf16 activation image
// (row r in fragment r / 16 as li = r % 16; its 8 elements at column 8k are chunk k = g + 4 c: F * 6144 + c * 512 + (16 g + li) * 8)
__global__ void gather_cls_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int nseq, int ntok, int img) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk
    if (i >= nseq * 48) return;
    const int s = i / 48, k = i % 48;
    const int64_t r = (int64_t)s * ntok;
    const int64_t off = img ? (r >> 4) * 6144 + (k >> 2) * 512 + (16 * (k & 3) + (int)(r & 15)) * 8 : r * 384 + 8 * k;
    *(u32x4*)(dst + (int64_t)s * 384 + 8 * k) = *(const u32x4*)(src + off);
}

}  // namespace

int hipt_attn_cls_launch(const void* qkv, void* out, float* probs, int B, int ntok, int heads, int dh, float scale, hipStream_t st) {
    HIPT_CHECK_ARG(dh == 64 && ntok > 0 && ntok <= 320, "attn_cls: head dim 64 and <= 320 tokens only (dh=%d ntok=%d)", dh, ntok);
    const int nbh = B * heads;
    hipLaunchKernelGGL(attn_cls_kernel, dim3((nbh + 3) / 4), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, probs, nbh, ntok, heads,
                       scale);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_attn_cls_probs_launch(const void* qkv, float* probs, int B, int ntok, int heads, int dh, float scale, int dtype, hipStream_t st) {
    HIPT_CHECK_ARG((dh == 64 || dh == 32) && ntok > 0 && ntok <= 320, "attn_cls_probs: head dim 32 / 64 and <= 320 tokens (dh=%d ntok=%d)", dh, ntok);
    const int nbh = B * heads;
    const dim3 grid((nbh + 3) / 4), blk(256);
#define CLSP(TT, DHH) hipLaunchKernelGGL((attn_cls_probs_kernel<TT, DHH>), grid, blk, 0, st, (const TT*)qkv, probs, nbh, ntok, heads, scale)
    if (dtype == HIPT_BF16) {
        if (dh == 64) CLSP(bf16_t, 64); else CLSP(bf16_t, 32);
    } else {
        if (dh == 64) CLSP(float, 64); else CLSP(float, 32);
    }
#undef CLSP
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_gather_cls_launch(const float* src, float* dst, int nseq, int64_t seq_stride, int D, hipStream_t st, int img) {
    HIPT_CHECK_ARG(D % 4 == 0, "gather_cls: D %% 4");
    HIPT_CHECK_ARG(!img || (D == 384 && seq_stride % D == 0), "gather_cls: image source needs D = 384 and whole rows");
    const int n = nseq * (D / 4);
    hipLaunchKernelGGL(gather_cls_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, dst, nseq, seq_stride, D, img);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_add_bf16_launch(float* out, const float* src, const void* y, int64_t n, hipStream_t st) {
    HIPT_CHECK_ARG(n % 8 == 0, "add_bf16: n %% 8 required");
    const int64_t n8 = n / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(add_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, src, (const bf16_t*)y, n8);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_layernorm_launch(const float* x, int64_t x_stride, const float* w, const float* b, void* out, int out_dtype,
                          int64_t out_stride, int rows, int D, float eps, hipStream_t st) {
    HIPT_CHECK_ARG(rows > 0 && D > 0 && D % 64 == 0 && D <= 64 * LN_MAX_E, "layernorm: D=%d must be a multiple of 64, <= %d",
                   D, 64 * LN_MAX_E);
    const dim3 grid((rows + 3) / 4), block(256);
    const bool v2 = D % 128 == 0 && x_stride % 2 == 0 && out_stride % 2 == 0 && ((uintptr_t)x % 8) == 0 &&
                    ((uintptr_t)out % 8) == 0;
    if (out_dtype == HIPT_F32) {
        if (v2)
            hipLaunchKernelGGL((ln_kernel<float, 2>), grid, block, 0, st, x, x_stride, w, b, (float*)out, out_stride, rows, D, eps);
        else
            hipLaunchKernelGGL((ln_kernel<float, 1>), grid, block, 0, st, x, x_stride, w, b, (float*)out, out_stride, rows, D, eps);
    } else if (out_dtype == HIPT_BF16) {
        if (v2)
            hipLaunchKernelGGL((ln_kernel<bf16_t, 2>), grid, block, 0, st, x, x_stride, w, b, (bf16_t*)out, out_stride, rows, D, eps);
        else
            hipLaunchKernelGGL((ln_kernel<bf16_t, 1>), grid, block, 0, st, x, x_stride, w, b, (bf16_t*)out, out_stride, rows, D, eps);
    } else {
        HIPT_CHECK_ARG(false, "layernorm: bad out dtype %d", out_dtype);
    }
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_cls_init_launch(float* x, const float* cls, const float* pos, int nseq, int ntok, int D, hipStream_t st) {
    const int n = nseq * D;
    hipLaunchKernelGGL(cls_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, cls, pos, nseq, ntok, D);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_cls_init_img_launch(float* x_img, void* xn_img, const float* cls, const float* pos, const float* ln_w, const float* ln_b, float ln_eps, int nseq,
                             int ntok, int D, hipStream_t st) {
    HIPT_CHECK_ARG(D == 384 && x_img && xn_img && ln_w && ln_b && nseq > 0 && ((int64_t)nseq * ntok) % 16 == 0, "cls_init_img: D = 384 and whole 16-row fragments");
    hipLaunchKernelGGL(cls_init_img_kernel, dim3(nseq), dim3(64), 0, st, x_img, (bf16_t*)xn_img, cls, pos, ln_w, ln_b, ln_eps, nseq, ntok);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_u8_normalize_launch(const void* src, int hwc, int64_t nimg, int64_t plane, void* dst, int dst_dtype, hipStream_t st) {
    HIPT_CHECK_ARG(src && dst && nimg > 0 && plane > 0 && plane % 16 == 0, "u8_normalize: empty input or W*H %% 16 != 0");
    HIPT_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "u8_normalize: 16-byte alignment required");
    const int64_t total = nimg * (plane / 16) * (hwc ? 1 : 3);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const dim3 grid((unsigned)blocks), block(256);
    const uint8_t* s8 = (const uint8_t*)src;
    if (dst_dtype == HIPT_F32) {
        if (hwc) hipLaunchKernelGGL((u8_norm_kernel<float, 1>), grid, block, 0, st, s8, (float*)dst, nimg, plane);
        else hipLaunchKernelGGL((u8_norm_kernel<float, 0>), grid, block, 0, st, s8, (float*)dst, nimg, plane);
    } else if (dst_dtype == HIPT_BF16) {
        if (hwc) hipLaunchKernelGGL((u8_norm_kernel<bf16_t, 1>), grid, block, 0, st, s8, (bf16_t*)dst, nimg, plane);
        else hipLaunchKernelGGL((u8_norm_kernel<bf16_t, 0>), grid, block, 0, st, s8, (bf16_t*)dst, nimg, plane);
    } else {
        HIPT_CHECK_ARG(false, "u8_normalize: bad dst dtype %d", dst_dtype);
    }
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_f32_to_bf16_launch(const float* in, void* out, int64_t n, hipStream_t st) {
    HIPT_CHECK_ARG(n % 8 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0,
                   "f32_to_bf16: n %% 8 and 16-byte alignment required");
    const int64_t n8 = n / 8;
    int64_t blocks = (n8 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, (bf16_t*)out, n8);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_gather_cls_bf16_launch(const void* src, void* dst, int nseq, int ntok, int D, hipStream_t st, int img) {
    HIPT_CHECK_ARG(D == 384, "gather_cls_bf16: D = 384 only");
    const int n = nseq * 48;
    hipLaunchKernelGGL(gather_cls_bf16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, nseq, ntok, img);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}
