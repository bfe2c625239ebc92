// Device helpers shared by the fused-MLP kernels (mlp.hip: generic persistent kernel; mlp_pipe.hip: the
// software-pipelined D = 384 kernel).
#pragma once
#include "common.h"

// GELU for the bf16 path.  nn.GELU() is the exact erf form x * Phi(x) (vision_transformer.py:89).  ocml erff
// costs ~36 VALU instructions per element, which at one wave per SIMD is as expensive as the MFMAs around
// it.  Here Phi(x) = 1 / (1 + exp2(x * q(x^2))) with q a degree-4 polynomial fitted (minimax on the GELU
// error, tools/fit_gelu.py) to -log2(e) * logit(Phi(x)) / x: |gelu error| <= 3.5e-6 for every finite fp32 x
// (the bf16 rounding of the result is 4e-3 relative), tails exact (q's leading term keeps the sign), and
// the arithmetic is all packed-fp32 (two elements per VALU instruction) plus one v_exp_f32 and one v_rcp_f32.
// (every multiply-add written out, fp contract off: all unrolled instances must round alike -- see pipe_common.h)
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
#pragma clang fp contract(off)
    const f32x2 t = x * x;
    f32x2 q = __builtin_elementwise_fma(t, f32x2{-3.228983431e-06f, -3.228983431e-06f}, f32x2{8.823808482e-05f, 8.823808482e-05f});
    q = __builtin_elementwise_fma(q, t, f32x2{3.602743489e-04f, 3.602743489e-04f});
    q = __builtin_elementwise_fma(q, t, f32x2{-1.052266864e-01f, -1.052266864e-01f});
    q = __builtin_elementwise_fma(q, t, f32x2{-2.302045392e+00f, -2.302045392e+00f});
    const f32x2 pw = x * q;
    f32x2 d;
    d[0] = __builtin_amdgcn_exp2f(pw[0]);
    d[1] = __builtin_amdgcn_exp2f(pw[1]);
    d = d + 1.0f;
    f32x2 r;
    r[0] = __builtin_amdgcn_rcpf(d[0]);
    r[1] = __builtin_amdgcn_rcpf(d[1]);
    return x * r;
}


// The same sigmoid form for ONE element, un-packed and with a degree-2 q (3 coefficients, x^2 clamped at 64 so that the positive
// leading term never flips the sign of the exponent): |gelu error| <= 2.6e-5 for every finite fp32 x (tools/fit_gelu.py 3).
// Why un-packed: v_pk_{mul,fma,add}_f32 do NOT run beside an MFMA -- tools/issue_mix_probe.hip: 16 packed FMAs in the gaps of
// four 32x32x16 MFMAs take the four MFMAs' 128 cycles PLUS 8 cycles each, with one or with two waves per SIMD, while plain v_fma_f32
// / v_exp_f32 issue in the 24 cycles per MFMA the matrix pipe leaves free.  mlp32.hip is compiled with -fno-slp-vectorize so
// that hipcc does not re-pack these.
__device__ __forceinline__ float gelu1(float x) {
#pragma clang fp contract(off)
    const float t = __builtin_fminf(x * x, 64.0f);
    float q = __builtin_fmaf(t, 1.014264505e-03f, -1.067757332e-01f);
    q = __builtin_fmaf(q, t, -2.301121329e+00f);
    const float d = __builtin_amdgcn_exp2f(x * q) + 1.0f;
    return x * __builtin_amdgcn_rcpf(d);
}

// gelu1 on pre-scaled data: xs = x / 8 in, gelu(x) / 8 out (mlp16.hip: its weight image holds W1 / 8 and 8 W2).  (x / 8)^2 clamped at 1 IS x^2 clamped
// at 64 -- and the clamp to [0, 1] is an output modifier of the multiply (v_mul_f32 ... clamp), not an instruction; the coefficients carry
// the powers of two, so every intermediate is the unscaled one times an exact power of two: the same bits as gelu1(x) / 8, 8 instead of
// 9 vector instructions per element.
__device__ __forceinline__ float gelu1s(float xs) {
#pragma clang fp contract(off)
    const float t = __builtin_fminf(__builtin_fmaxf(xs * xs, 0.0f), 1.0f);
    float q = __builtin_fmaf(t, 1.014264505e-03f * 32768.0f, -1.067757332e-01f * 512.0f);
    q = __builtin_fmaf(q, t, -2.301121329e+00f * 8.0f);
    const float d = __builtin_amdgcn_exp2f(xs * q) + 1.0f;
    return xs * __builtin_amdgcn_rcpf(d);
}

template <int NCH>
__device__ __forceinline__ void ln_rows(f32x4 (&v)[NCH][2], const float* gam, const float* bet, int K, float eps, int g,
                                        u32x4 (&out)[NCH]) {
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[c][0][e] + v[c][1][e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)K;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = v[c][0][e] - mean, b = v[c][1][e] - mean;
            q = __builtin_fmaf(a, a, q);
            q = __builtin_fmaf(b, b, q);
        }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q / (float)K + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int k0 = (g + 4 * c) * 8;
        const f32x4 g0 = *(const f32x4*)(gam + k0), g1 = *(const f32x4*)(gam + k0 + 4);
        const f32x4 b0 = *(const f32x4*)(bet + k0), b1 = *(const f32x4*)(bet + k0 + 4);
        f32x4 y0, y1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            y0[e] = __builtin_fmaf((v[c][0][e] - mean) * rstd, g0[e], b0[e]);
            y1[e] = __builtin_fmaf((v[c][1][e] - mean) * rstd, g1[e], b1[e]);
        }
        u32x4 o;
        o[0] = pack_bf16x2(y0[0], y0[1]);
        o[1] = pack_bf16x2(y0[2], y0[3]);
        o[2] = pack_bf16x2(y1[0], y1[1]);
        o[3] = pack_bf16x2(y1[2], y1[3]);
        out[c] = o;
    }
}

