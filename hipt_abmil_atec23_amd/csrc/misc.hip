// LayerNorm (vision_transformer.py:138,142,195), [CLS] row initialisation (:240-241,244) and the
// fp32 -> bf16 conversion used by the bf16 mode.  All HBM-bound: one wave per row, coalesced reads,
// wavefront (64-lane) shuffle reductions, fp32 statistics with the two-pass (mean, then centred
// variance) form torch uses.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int LN_MAX_E = 32;  // D <= 2048

template <typename TO>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, int64_t x_stride,
                                                 const float* __restrict__ w, const float* __restrict__ b,
                                                 TO* __restrict__ out, int64_t out_stride, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * x_stride;
    const int E = D >> 6;
    float v[LN_MAX_E];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_E; ++i)
        if (i < E) {
            v[i] = xr[lane + 64 * i];
            s += v[i];
        }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_E; ++i)
        if (i < E) {
            const float c = v[i] - mean;
            q += c * c;
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    TO* orow = out + (int64_t)row * out_stride;
#pragma unroll
    for (int i = 0; i < LN_MAX_E; ++i)
        if (i < E) {
            const int d = lane + 64 * i;
            orow[d] = (TO)((v[i] - mean) * rstd * w[d] + b[d]);
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

}  // namespace

int hipt_layernorm_launch(const float* x, int64_t x_stride, const float* w, const float* b, void* out, int out_dtype,
                          int64_t out_stride, int rows, int D, float eps, hipStream_t st) {
    HIPT_CHECK_ARG(rows > 0 && D > 0 && D % 64 == 0 && D <= 64 * LN_MAX_E, "layernorm: D=%d must be a multiple of 64, <= %d",
                   D, 64 * LN_MAX_E);
    const dim3 grid((rows + 3) / 4), block(256);
    if (out_dtype == HIPT_F32)
        hipLaunchKernelGGL(ln_kernel<float>, grid, block, 0, st, x, x_stride, w, b, (float*)out, out_stride, rows, D, eps);
    else if (out_dtype == HIPT_BF16)
        hipLaunchKernelGGL(ln_kernel<bf16_t>, grid, block, 0, st, x, x_stride, w, b, (bf16_t*)out, out_stride, rows, D, eps);
    else
        HIPT_CHECK_ARG(false, "layernorm: bad out dtype %d", out_dtype);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_cls_init_launch(float* x, const float* cls, const float* pos, int nseq, int ntok, int D, hipStream_t st) {
    const int n = nseq * D;
    hipLaunchKernelGGL(cls_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, cls, pos, nseq, ntok, D);
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
