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

}  // namespace

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
