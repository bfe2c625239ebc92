// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16, LDS tiles of 128-byte rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hipt_abmil.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))

// ---- per-type traits: a "chunk" is 16 bytes = one lane's MFMA operand fragment -----------
// One LDS tile row is 128 bytes = 8 chunks: 64 bf16 or 32 fp32 along K.
// mma16(acc, a, b): acc[16x16] += A'[16 x kc] * B'[16 x kc]^T where lane l supplies row (l & 15)
// of A' in `a` and row (l & 15) of B' in `b`, both for K-slot group (l >> 4).
//   bf16: one v_mfma_f32_16x16x32_bf16 (group g carries k = 8g..8g+7).
//   fp32: four v_mfma_f32_16x16x4_f32; step j uses element j of both fragments, i.e. group g
//         carries k = 4g+j.  Any K permutation is legal as long as A and B agree.
// Result layout (both): acc[i] = C[row 4*(l>>4)+i of A'-rows][col (l&15) of B'-rows].
template <typename T> struct Tr;

template <> struct Tr<float> {
    static constexpr int EPC = 4;    // elements per 16-byte chunk
    static constexpr int KB = 32;    // elements per 128-byte LDS row
    static constexpr int DT = HIPT_F32;
    static __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b) {
        // NB: bit-cast the whole vector, not a[j]: __builtin_bit_cast on a vector-element lvalue
        // reads element 0 (observed with ROCm 7.2 clang)
        const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f(float v) { return v; }
    static __device__ __forceinline__ float from_f(float v) { return v; }
};

template <> struct Tr<bf16_t> {
    static constexpr int EPC = 8;
    static constexpr int KB = 64;
    static constexpr int DT = HIPT_BF16;
    static __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float to_f(bf16_t v) { return (float)v; }
    static __device__ __forceinline__ bf16_t from_f(float v) { return (bf16_t)v; }
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 p;
    p[0] = (bf16_t)lo;
    p[1] = (bf16_t)hi;
    return __builtin_bit_cast(uint32_t, p);
}

// store 4 consecutive outputs (columns n..n+3 of one row) as T
template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const f32x4& v) {
    u32x2 o;
    o[0] = pack_bf16x2(v[0], v[1]);
    o[1] = pack_bf16x2(v[2], v[3]);
    *(u32x2*)p = o;
}

__device__ __forceinline__ float gelu_erf(float x) {  // nn.GELU() default (exact erf)
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// async global -> LDS, 16 bytes per lane; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (LDS_AS void*)lds_wave_base, 16, 0, 0);
}
// ---- "invisible" LDS accesses -----------------------------------------------------------------------
// While LDS-DMA (global_load_lds) is in flight, hipcc cannot prove that a C++-level LDS access does not
// alias the DMA destination and inserts s_waitcnt vmcnt(0) in front of it — draining the whole DMA ring.
// Inside ring loops every LDS access therefore goes through these helpers (the compiler sees no LDS op);
// each carries its own lgkmcnt(0), so they are for the non-hot accesses (biases, tiny exchange buffers).
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(LDS_AS const char*)p; }
__device__ __forceinline__ f32x4 lds_ld128(uint32_t a) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ float lds_ld32(uint32_t a) {
    float v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ u32x2 lds_ld_tr64(uint32_t a) {  // ds_read_b64_tr_b16: EXEC must be all ones
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
__device__ __forceinline__ void lds_st32(uint32_t a, float v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_st64(uint32_t a, u32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(v) : "memory"); }

__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// XCD-aware bijective remap of a linear block id: blocks b and b+8 share an XCD (and its L2),
// so give each XCD a contiguous range of logical tiles (cdna guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, j = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

// host side -------------------------------------------------------------------------------
void hipt_set_error(const char* fmt, ...);

// The library reads SIX environment switches, all for debugging / A-B runs (README.md):
//   HIPT_GENERIC=1        every operator takes its generic kernel (no streaming / packed-weight kernels)
//   HIPT_NO_IMG=1         no activation images / head-major qkv between the streaming kernels
//   HIPT_NO_PRUNE=1       the last ViT-256 block runs in full instead of for the [CLS] rows only
//   HIPT_NO_FUSED_ATTN=1  LayerNorm-chained blocks run QKV GEMM + attention as two kernels instead of the fused one
//   HIPT_NO_PROJ_FOLD=1   the attention block's output projection runs as its own kernel instead of at the head of the fused MLP's tiles
//   HIPT_NO_EMBED_LN=1    the patch embedding writes row-major tokens only; the first block applies its own LayerNorm-1 (LN-in-GEMM + two-kernel attention)
// read per call (cheap: host side, a handful of calls per forward), so a test may flip them inside one process.
#include <stdlib.h>
inline bool hipt_env_on(const char* name) {
    const char* v = getenv(name);
    return v != nullptr && v[0] != '\0' && v[0] != '0';
}
inline bool hipt_generic_only() { return hipt_env_on("HIPT_GENERIC"); }

// hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) is a PER-DEVICE setting and a process may drive several GPUs
// (HIPT_4K's device256 != device4k placement, a module on cuda:1 while cuda:0 is current): launchers cache it per
// (kernel family, device).  hipt_cur_device(): ordinal of the calling thread's current device, -1 on failure.
constexpr int HIPT_MAX_DEV = 64;
struct DevOnce {
    bool done[HIPT_MAX_DEV] = {};
    int ncu[HIPT_MAX_DEV] = {};
};
inline int hipt_cur_device() {
    int d = 0;
    return (hipGetDevice(&d) == hipSuccess && d >= 0 && d < HIPT_MAX_DEV) ? d : -1;
}
// In-kernel time stamps and the host code that reads them back (hipMalloc, hipStreamSynchronize, hipMemcpy) exist only in
// diagnostic builds (make DEBUG_STAMPS=1 -> libhipt_abmil_dbg.so).
#ifdef HIPT_DEBUG_STAMPS
#define HIPT_STAMPS_ON(ptr) ((ptr) != nullptr)
#else
#define HIPT_STAMPS_ON(ptr) false
#endif
#define HIPT_CUR_DEVICE(dev)                                              \
    const int dev = hipt_cur_device();                                    \
    if (dev < 0) {                                                        \
        hipt_set_error("%s:%d: hipGetDevice failed", __FILE__, __LINE__); \
        return HIPT_E_LAUNCH;                                             \
    }
#define HIPT_CHECK_ARG(cond, ...)                \
    do {                                         \
        if (!(cond)) {                           \
            hipt_set_error(__VA_ARGS__);         \
            return HIPT_E_BADARG;                \
        }                                        \
    } while (0)
#define HIPT_CHECK_LAUNCH()                                                        \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            hipt_set_error("%s:%d: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return HIPT_E_LAUNCH;                                                  \
        }                                                                          \
    } while (0)
