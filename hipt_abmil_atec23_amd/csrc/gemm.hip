// out = epilogue(A[M,K] @ W[N,K]^T + bias): the nn.Linear / Conv2d-as-GEMM workhorse of the ViT path
// (reference call sites: HIPT_4K/vision_transformer.py:93-95,114,116,165; vision_transformer4k.py:169).
//
// gfx950 design: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 4x4 MFMA 16x16 tiles), K streamed in 128-byte slabs (64 bf16 / 32 fp32) through a 2-stage LDS
// ring filled by LDS-DMA (global_load_lds_dwordx4).  The LDS image is lane-linear per DMA
// instruction, so the bank-conflict swizzle (16-byte chunk index ^= (row>>1)&7) is applied to the
// per-lane SOURCE address and again on the ds_read_b128 side.  Operands are fed "swapped"
// (weights as the MFMA A operand) so that each lane ends up with 4 consecutive output columns of
// one row and the epilogue stores 8/16 bytes per lane without an LDS round trip.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int STAGE_BYTES = (BM + BN) * 128;  // 32 KiB
constexpr int GEMM_LDS = 2 * STAGE_BYTES;     // 64 KiB -> 2 workgroups per CU

template <typename T, int ALOAD>
struct ALoader {
    // per-lane state for the 4 LDS-DMA instructions this wave issues per K slab for the A tile
    const T* base[4];
    int64_t cs, rs;
    __device__ __forceinline__ void init(const GemmParams& p, int m0, int wave, int lane) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = (wave * 4 + q) * 8 + (lane >> 3);
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            if constexpr (ALOAD == ALOAD_PLAIN) {
                base[q] = (const T*)p.A + (int64_t)m * p.lda;
            } else {
                const int tps = p.im_nty * p.im_ntx;
                const int b = p.im_seq0 + m / tps, t = m % tps;
                const int ty = t / p.im_ntx, tx = t % p.im_ntx;
                const int gsz = p.im.grid_w * p.im.grid_h;
                const int bi = b / gsz, s = b % gsz;
                const int p1 = s / p.im.grid_h, p2 = s % p.im.grid_h;
                base[q] = (const T*)p.A + (int64_t)bi * p.im.batch_stride +
                          (int64_t)(p1 * p.im.patch_h + ty * 16) * p.im.row_stride + p2 * p.im.patch_w + tx * 16;
            }
        }
        cs = p.im.chan_stride;
        rs = p.im.row_stride;
    }
    // source of logical chunk `kc` (16 bytes, global chunk index along K) of instruction q's row
    __device__ __forceinline__ const T* src(int q, int kc) const {
        if constexpr (ALOAD == ALOAD_PLAIN) {
            return base[q] + kc * Tr<T>::EPC;
        } else {
            const int k = kc * Tr<T>::EPC;  // k = c*256 + ky*16 + kx
            return base[q] + (int64_t)(k >> 8) * cs + (int64_t)((k >> 4) & 15) * rs + (k & 15);
        }
    }
};

template <typename T, int FLAGS>
__device__ __forceinline__ void epilogue(const GemmParams& p, int m, int n, f32x4 v) {
    if (p.bias) v += *(const f32x4*)(p.bias + n);
    if constexpr (FLAGS & HIPT_EPI_GELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gelu_erf(v[i]);
    }
    if constexpr (FLAGS & HIPT_EPI_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
    }
    int64_t orow = m;
    if constexpr (FLAGS & EPI_ROWMAP) {  // token rows: skip the [CLS] slot of every sequence, add pos
        const int s = m / p.rows_per_seq, t = m % p.rows_per_seq;
        orow = (int64_t)s * (p.rows_per_seq + 1) + 1 + t;
        v += *(const f32x4*)(p.pos + (int64_t)(t + 1) * p.N + n);
    }
    if constexpr (FLAGS & HIPT_EPI_RESID) v += *(const f32x4*)(p.resid + orow * p.ldc + n);
    if constexpr (FLAGS & HIPT_EPI_OUT_F32)
        store4<float>((float*)p.out + orow * p.ldc + n, v);
    else
        store4<T>((T*)p.out + orow * p.ldc + n, v);
}

template <typename T, int ALOAD, int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;

    // ---- LDS-DMA source addressing ----
    ALoader<T, ALOAD> al;
    al.init(p, m0, wave, lane);
    const T* wbase[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int n = n0 + (wave * 4 + q) * 8 + (lane >> 3);
        n = n < p.N ? n : p.N - 1;
        wbase[q] = (const T*)p.W + (int64_t)n * p.ldw;
    }
    // instruction q of this wave fills rows (wave*4+q)*8 .. +7; lane -> (row = lane>>3, physical
    // chunk = lane&7); the logical chunk it must fetch is phys ^ ((row>>1)&7), row&15 pattern only
    int lchunk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + (lane >> 3);
        lchunk[q] = (lane & 7) ^ ((r >> 1) & 7);
    }
    auto stage = [&](int s, int kt) {
        char* sa = smem + s * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(al.src(q, kt * 8 + lchunk[q]), sa + (wave * 4 + q) * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            glds16(wbase[q] + (kt * 8 + lchunk[q]) * Tr<T>::EPC, sa + BM * 128 + (wave * 4 + q) * 1024);
    };

    // ---- fragment read offsets (bytes within a tile): row (lane&15), chunk ((lane>>4)+4ks) swizzled
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        foff[ks] = (lane & 15) * 128 + ((((lane >> 4) + 4 * ks) ^ ((lane >> 1) & 7)) << 4);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / Tr<T>::KB;
    stage(0, 0);
    wait_vm0();
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sa = smem + cur * STAGE_BYTES + wm * 64 * 128;
        const char* sw = smem + cur * STAGE_BYTES + BM * 128 + wn * 64 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const u32x4*)(sa + i * 16 * 128 + foff[ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *(const u32x4*)(sw + j * 16 * 128 + foff[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Tr<T>::mma16(acc[i][j], wf[j], af[i]);
        }
        wait_vm0();
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3] ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
            if (n < p.N) epilogue<T, FLAGS>(p, m, n, acc[i][j]);
        }
    }
}

template <typename T, int ALOAD, int FLAGS>
int launch(const GemmParams& p, hipStream_t st) {
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    static bool attr_set = false;  // 64 KiB dynamic LDS needs the opt-in once per kernel
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_kernel<T, ALOAD, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GEMM_LDS) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(gemm) failed");
            return HIPT_E_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_kernel<T, ALOAD, FLAGS>), dim3(tiles), dim3(256), GEMM_LDS, st, p);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

template <typename T>
int dispatch(const GemmParams& p, int aload, int flags, hipStream_t st) {
#define CASE(AL, FL) \
    if (aload == AL && flags == (FL)) return launch<T, AL, (FL)>(p, st);
    CASE(ALOAD_PLAIN, 0)
    CASE(ALOAD_PLAIN, HIPT_EPI_GELU)
    CASE(ALOAD_PLAIN, HIPT_EPI_RELU)
    CASE(ALOAD_PLAIN, HIPT_EPI_OUT_F32)
    CASE(ALOAD_PLAIN, HIPT_EPI_RELU | HIPT_EPI_OUT_F32)
    CASE(ALOAD_PLAIN, HIPT_EPI_GELU | HIPT_EPI_OUT_F32)
    CASE(ALOAD_PLAIN, HIPT_EPI_RESID | HIPT_EPI_OUT_F32)
    CASE(ALOAD_PLAIN, HIPT_EPI_GELU | HIPT_EPI_OUT_F32 | EPI_ROWMAP)
    CASE(ALOAD_IM2COL, HIPT_EPI_OUT_F32 | EPI_ROWMAP)
#undef CASE
    hipt_set_error("gemm: unsupported loader/epilogue combination (aload=%d flags=%d)", aload, flags);
    return HIPT_E_UNSUPPORTED;
}

}  // namespace

int hipt_gemm_launch(const GemmParams& p, int dtype, int aload, int flags, hipStream_t st) {
    const int kb = dtype == HIPT_F32 ? 32 : 64;
    HIPT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    HIPT_CHECK_ARG(p.K % kb == 0, "gemm: K=%d must be a multiple of %d", p.K, kb);
    HIPT_CHECK_ARG(p.N % 4 == 0, "gemm: N=%d must be a multiple of 4", p.N);
    HIPT_CHECK_ARG(p.ldc % 4 == 0, "gemm: ldc=%lld must be a multiple of 4", (long long)p.ldc);
    const int esz = dtype == HIPT_F32 ? 4 : 2;
    HIPT_CHECK_ARG(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0 && ((uintptr_t)p.out % 16) == 0,
                   "gemm: A/W/out must be 16-byte aligned");
    if (aload == ALOAD_PLAIN)
        HIPT_CHECK_ARG((p.lda * esz) % 16 == 0, "gemm: lda rows must be 16-byte multiples");
    else
        HIPT_CHECK_ARG((p.im.row_stride * esz) % 16 == 0 && (p.im.chan_stride * esz) % 16 == 0 &&
                           (p.im.batch_stride * esz) % 16 == 0 && p.K == 768,
                       "gemm/im2col: image strides must be 16-byte multiples and K == 768");
    HIPT_CHECK_ARG((p.ldw * esz) % 16 == 0, "gemm: ldw rows must be 16-byte multiples");
    if (dtype == HIPT_F32) return dispatch<float>(p, aload, flags, st);
    if (dtype == HIPT_BF16) return dispatch<bf16_t>(p, aload, flags, st);
    hipt_set_error("gemm: bad dtype %d", dtype);
    return HIPT_E_BADARG;
}
