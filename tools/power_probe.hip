// Bare bf16 MFMA loops as a shared library (round 6): tools/power_legs.py launches them from Python while a thread of the
// same process samples the board's hwmon node, so that the power and the clock a loop holds are MEASURED beside each other.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/power_probe.hip -o tools/probe_bin/libpower_probe.so
// One workgroup of 256 threads per CU (one wave per SIMD), operands in registers (random bf16 in [1, 2) with random signs), eight
// independent 16x16 accumulators or two 32x32 ones; per iteration 4 x 32x32x16 (32 cycles each: 128 matrix-pipe cycles, 131 072 FLOP per SIMD) or
// 16 x 16x16x32 on sixteen accumulators (16 cycles each: 256 cycles, 262 144 FLOP; with eight accumulators the loop ran at 78 % of the pipe: a
// dependent 16x16x32 needs more than eight issue slots of distance) -- pp_flop_per_iter() tells the caller.  out[2 * wave_global + {0, 1}] = (shader cycles, 100 MHz ticks) around the loop: the in-kernel clock (MI355X_MICROARCH.md,
// DVFS give-back item 6).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void pp_kernel(int iters, unsigned long long* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 w[4], b;
    for (int i = 0; i < 4; ++i) b[i] = (((uint32_t)((threadIdx.x + 256 * blockIdx.x) * 2246822519u + i * 3266489917u)) & 0x807f807fu) | 0x3c003c00u;
    for (int i = 0; i < 4; ++i) w[i] = b ^ (uint32_t)(i * 0x00010001u);
    f32x16 a0 = {}, a1 = {};
    f32x4 c[16] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // (a scheduling barrier after every MFMA, as tools/mfma_shape_probe.hip: left alone, hipcc re-orders the 16x16x32 MFMAs into
        //  dependent pairs -- 33 cycles per MFMA instead of 16)
        if constexpr (SHAPE == 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16& acc = (j & 1) ? a1 : a0;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[j]), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[(j >> 2) & 3]), __builtin_bit_cast(bf16x8, b), c[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e];
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    if (lane == 0) {
        out[2 * (blockIdx.x * 4 + wave)] = t1 - t0;
        out[2 * (blockIdx.x * 4 + wave) + 1] = r1 - r0;
    }
    if (s == 123.456f) out[0] = 0;
}

// shape: 16 or 32; out: device buffer of 2 * 4 * nblocks uint64 (or more).  Enqueues only.  FLOP of a launch = nblocks * 4 * iters * pp_flop_per_iter(shape).
extern "C" double pp_flop_per_iter(int shape) { return shape == 32 ? 131072.0 : 262144.0; }
extern "C" int pp_launch(int shape, int nblocks, int iters, void* out, void* stream) {
    if (shape == 32) hipLaunchKernelGGL(pp_kernel<32>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)out);
    else hipLaunchKernelGGL(pp_kernel<16>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, iters, (unsigned long long*)out);
    return (int)hipGetLastError();
}
