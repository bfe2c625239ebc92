// Building blocks of the software-pipelined A-stationary kernels (mlp_pipe.hip, seqgemm_pipe.hip): compile-time
// loops, inline-asm LDS reads with counted waits, LayerNorm into MFMA operand fragments.
#pragma once
#include <type_traits>

#include "common.h"

template <int I, int N, class F> __device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

#define DSR128(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define DSR64(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
// four / eight LDS reads AND their wait in ONE statement: the outputs are valid when the statement ends, so the
// compiler may do what it likes with them (the split form -- reads, then a counted wait -- is only safe where
// nothing makes hipcc copy or re-use the destinations in between: tools/audit_asm_reads.py checks the .s)
#define DSR128X4_WAIT(d0, d1, d2, d3, addr, o0, o1, o2, o3)                                                      \
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\t" \
                 "ds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"                                           \
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)                                                     \
                 : "v"(addr), "n"(o0), "n"(o1), "n"(o2), "n"(o3))
#define LGKM(n)                                                 \
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory"); \
    __builtin_amdgcn_sched_barrier(0)

// LayerNorm of the 16-row fragment a lane quartet holds (lanes sharing lane&15 own one row), result packed as MFMA
// operand chunks.  gamma / beta come from LDS through asm reads with immediate offsets off ONE address register:
// written as C++ loads, hipcc hoists the 48 per-chunk addresses out of the tile loop and spills every one of them.
// gaddr = LDS byte address of gamma + 32 g;  beta sits D floats behind gamma.
// Every multiply-add is written out (fp contract off): left to hipcc, the unrolled instances of this function get
// DIFFERENT contractions (mul+add here, fma there), i.e. a row's result would depend on which fragment of a tile
// it lands in -- the outputs must not depend on how rows are batched.
template <int D, int NCH>
__device__ __forceinline__ void ln_rows_lds(f32x4 (&v)[NCH][2], uint32_t gaddr, float eps, u32x4 (&out)[NCH]) {
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[c][0][e] + v[c][1][e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / D);
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
    const float rstd = 1.0f / sqrtf(q * (1.0f / D) + eps);
    sfor<0, NCH>([&](auto C_) __attribute__((always_inline)) {
#pragma clang fp contract(off)
        constexpr int c = decltype(C_)::value;
        f32x4 g0, g1, b0, b1;
        const uint32_t ga = gaddr;
        DSR128X4_WAIT(g0, g1, b0, b1, ga, c * 128, c * 128 + 16, c * 128 + D * 4, c * 128 + D * 4 + 16);
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
    });
}
