// out = epilogue(A[M,K] @ W[N,K]^T + bias): the nn.Linear / Conv2d-as-GEMM workhorse of the ViT path
// (reference call sites: HIPT_4K/vision_transformer.py:93-95,114,116,165; vision_transformer4k.py:169).
//
// gfx950 design.  The token matrix of a region is 256 patches x 257 tokens; 257 is prime, so any
// power-of-two M tile leaves a ragged last round on 256 CUs (128-row tiles: 1542 tiles of the proj
// GEMM on 512 slots = 4 rounds for 3.01 rounds of work).  The M tile is therefore ONE SEQUENCE:
// `rpt` rows (257, or 256 for the patch-embedding GEMM) padded to 17 MFMA row fragments (272 rows,
// 5.5 % padding), times 128 output columns -> grid = sequences x N/128, an exact multiple of the CU
// count for a 256-patch region, one 512-thread workgroup (8 waves as 2(M) x 4(N)) per CU.
// K streams in 128-byte slabs (64 bf16 / 32 fp32) through a 3-stage LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4) that stays in flight across the single raw s_barrier per slab (counted
// s_waitcnt vmcnt, never 0 inside the loop).  The LDS image is lane-linear per DMA instruction, so the
// bank-conflict swizzle (16-byte chunk index ^= (row>>1)&7) is applied to the per-lane SOURCE address
// and again on the ds_read_b128 side.  Operands are fed "swapped" (weights as the MFMA A operand) so
// each lane ends with 4 consecutive output columns of one row: 8/16-byte epilogue stores, no LDS hop.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MF = 17;                  // MFMA row fragments per tile (272 rows)
constexpr int TROWS = MF * 16;          // 272
constexpr int BN = 128;
constexpr int A_BYTES = TROWS * 128;    // 34816
constexpr int STAGE_BYTES = A_BYTES + BN * 128;  // 51200
constexpr int NSTAGE = 3;
constexpr int GEMM_LDS = NSTAGE * STAGE_BYTES;   // 153600 B -> one workgroup per CU
constexpr int A_INSTR = TROWS / 8;      // 34 DMA instructions of 1 KiB per A slab

template <typename T, int ALOAD>
struct ALoader {
    const T* base[5];  // this wave issues A instructions q*8 + wave, q = 0..4 (the 5th only for wave < 2)
    int64_t cs, rs;
    __device__ __forceinline__ void init(const GemmParams& p, int row0, int nrows, int wave, int lane) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            int r = (q * 8 + wave) * 8 + (lane >> 3);
            r = r < nrows ? r : nrows - 1;  // padding rows re-read the last valid row (never stored)
            const int m = row0 + r;
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
    // source of logical chunk `kc` (16 bytes, global chunk index along K) of instruction slot q
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

template <int N> __device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int ALOAD, int FLAGS>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // 2 (M) x 4 (N)
    const int li = lane & 15, g = lane >> 4;

    // consecutive blocks on one XCD walk the N tiles of the same sequence: A slab re-reads hit that L2
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (tile / tiles_n) * p.rpt, n0 = (tile % tiles_n) * BN;
    int nrows = p.M - row0;
    nrows = nrows < p.rpt ? nrows : p.rpt;

    // ---- LDS-DMA source addressing ----
    ALoader<T, ALOAD> al;
    al.init(p, row0, nrows, wave, lane);
    const T* wbase[2];
    int wchunk[2], achunk[5];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (q * 8 + wave) * 8 + (lane >> 3);
        int n = n0 + r;
        n = n < p.N ? n : p.N - 1;
        wbase[q] = (const T*)p.W + (int64_t)n * p.ldw;
        wchunk[q] = (lane & 7) ^ ((r >> 1) & 7);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int r = (q * 8 + wave) * 8 + (lane >> 3);
        achunk[q] = (lane & 7) ^ ((r >> 1) & 7);
    }
    const bool five = wave < (A_INSTR - 32);  // waves 0,1 issue a 5th A instruction (34 = 4*8 + 2)
    auto stage = [&](int s, int kt) {
        char* sa = smem + s * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(al.src(q, kt * 8 + achunk[q]), sa + (q * 8 + wave) * 1024);
        if (five) glds16(al.src(4, kt * 8 + achunk[4]), sa + (32 + wave) * 1024);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            glds16(wbase[q] + (kt * 8 + wchunk[q]) * Tr<T>::EPC, sa + A_BYTES + (q * 8 + wave) * 1024);
    };

    // ---- fragment read offsets (bytes within a tile): row li, chunk (g + 4ks) swizzled ----
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = li * 128 + (((g + 4 * ks) ^ ((lane >> 1) & 7)) << 4);

    // wave row wm owns row fragments [wm*9, wm*9 + nf), nf = 9 / 8
    constexpr int NF = 9;
    const int f0 = wm * NF;
    f32x4 acc[NF][2];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / Tr<T>::KB;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    for (int kt = 0; kt < nk; ++kt) {
        // slab kt has landed once at most the loads of slab kt+1 are outstanding (6 or 7 per wave)
        if (kt + 1 < nk) {
            if (five) wait_vm<7>(); else wait_vm<6>();
        } else {
            wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();  // everyone's slab kt is visible; everyone is done reading slab kt-1
        if (kt + 2 < nk) stage((kt + 2) % NSTAGE, kt + 2);  // overwrites the buffer of slab kt-1
        const char* sa = smem + (kt % NSTAGE) * STAGE_BYTES + f0 * 16 * 128;
        const char* sw = smem + (kt % NSTAGE) * STAGE_BYTES + A_BYTES + wn * 32 * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 wf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = *(const u32x4*)(sw + j * 16 * 128 + foff[ks]);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                if (i == NF - 1 && wm == 1) break;  // second wave row has 8 fragments (17 = 9 + 8)
                const u32x4 af = *(const u32x4*)(sa + i * 16 * 128 + foff[ks]);
#pragma unroll
                for (int j = 0; j < 2; ++j) Tr<T>::mma16(acc[i][j], wf[j], af);
            }
        }
    }

    // ---- epilogue: lane holds C[row = frag*16 + li][n = n0 + wn*32 + j*16 + 4g + 0..3] ----
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        if (i == NF - 1 && wm == 1) break;
        const int r = (f0 + i) * 16 + li;
        if (r >= nrows) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + 4 * g;
            if (n < p.N) epilogue<T, FLAGS>(p, row0 + r, n, acc[i][j]);
        }
    }
}

// ---- small M (a few hundred rows: ONE 256 x 256 patch through ViT-256 is 257, BASELINE configs[1]; the second-level ViT of a region;
// the [CLS] rows of a launch) ----
// The sequence-tiled kernel above puts such a problem on N / 128 workgroups -- 3 for the proj / fc2 Linears of ViT-256, 9 for QKV:
// a fp32 one-patch forward spent 5.8 ms on nine CUs.  Here a one-WAVE workgroup owns one 16-row fragment x 32 output columns and
// streams both operands straight from L2 (the whole problem is a few MB: no LDS, no barriers): ceil(M / 16) x ceil(N / 32)
// workgroups -- 204 to 816 for ViT-256's Linears at 257 rows, a CU's vector-memory path per wave (four waves on one CU shared it:
// fc2 took 18.5 us on 51 CUs, bound by the L1 fill rate).
//   k order.  One mma16 consumes KC = 4 EPC k values, lane group g supplying one 16-byte chunk per operand row.  WHICH k values a
// chunk holds is free as long as both operands agree, so a PAIR of steps covers 128 consecutive bytes of every row -- lane group g
// owns bytes [32 g, 32 g + 32) of them, first half for the even step, second half for the odd one: the two requests of a pair touch
// the same 16 (or 32) cache lines back to back.  (With 64 bytes of a row per step, the second half of each line was fetched again
// a step later, after the ring's other requests had pushed it out of the 32 KiB L1.)
constexpr int SMALL_M = 1088;  // at most four 272-row sequences take this kernel
constexpr int SMALL_RING = 6;  // k-steps of operands in flight per wave (6 divides the step count of every ViT Linear)

// element offset of step s inside an operand row, for the lane whose pointer already holds + 2 EPC g  (paired order, see above)
template <int EPC> __device__ __forceinline__ constexpr int pair_off(int s) { return (s >> 1) * 8 * EPC + (s & 1) * EPC; }

// AIMG (bf16, K = 384, EVEN): A is a bf16 activation image (kernels.h) and GEMM row r is image row r * p.a_row_step -- the [CLS] rows
// of the sequences, read where the fused MLP left them (the side GEMM of the fused attention unit; a gather launch before it otherwise).
// Eight consecutive columns from a multiple of 8 are 16 consecutive bytes of the image too: fragment (R >> 4) * 6144 elements, row R & 15
// at + 8 (R & 15), chunk m = col / 8 at (m >> 2) * 512 + ((m >> 1) & 1) * 256 + (m & 1) * 128; in the paired k order chunk m of step s,
// lane group g is 8 (s >> 1) + 2 g + (s & 1): step part (s >> 1) * 1024 + (s & 1) * 128, lane part (g >> 1) * 512 + (g & 1) * 256.
// AI2C (EVEN, K = 768): the patch embedding of a few patches -- A is the image tensor in the compute dtype and row m is token m of
// the call (the sequence-tiled kernel's im2col addressing: k = 256 c + 16 ky + kx -> channel c, pixel (16 ty + ky, 16 tx + kx) of the
// patch); a 16-byte chunk never crosses a pixel row.  One patch is 256 tokens: three workgroups of the tiled kernel (115 us in fp32), 192 here.
// k order: ASCENDING in steps of KC consecutive values, no pairing, no staggered start -- the k sets and the order of the tiled kernel's
// MFMAs (and of embed32.hip's), so that the tokens are the same bits whichever of the three kernels a call's size selects; and nothing is
// lost: the 16 rows of a wave are the 16 tokens of one token row, whose 64-byte pixel runs are consecutive in the image.
// ASC (EVEN): the same ascending order for plain / image rows (GemmParams::asc): one-row-per-sequence GEMMs whose larger calls run on the tiled kernel.
template <typename T, int FLAGS, bool EVEN, bool AIMG = false, bool AI2C = false, bool ASC = false>
__global__ __launch_bounds__(64) void gemm_small_kernel(const GemmParams p) {
    const int lane = threadIdx.x;
    const int li = lane & 15, g = lane >> 4;
    const int tiles_n = (p.N + 31) / 32;
    const int row0 = (blockIdx.x / tiles_n) * 16, n0 = (blockIdx.x % tiles_n) * 32;
    constexpr int EPC = Tr<T>::EPC, KC = 4 * EPC;
    int r = row0 + li;
    r = r < p.M ? r : p.M - 1;                      // (padding rows re-read the last valid row: never stored)
    const T* ap = (const T*)p.A + (int64_t)r * p.lda + g * (EVEN && !ASC ? 2 * EPC : EPC);
    if constexpr (AIMG) {
        const int64_t R = (int64_t)r * p.a_row_step;
        // (ascending order: chunk m = 4 kk + g of step kk -> lane part (g >> 1) * 256 + (g & 1) * 128, step part kk * 512)
        ap = (const T*)p.A + (R >> 4) * (16 * 384) + (int)(R & 15) * 8 + (ASC ? (g >> 1) * 256 + (g & 1) * 128 : (g >> 1) * 512 + (g & 1) * 256);
    }
    const int i2c_rs = (int)p.im.row_stride, i2c_cs = (int)p.im.chan_stride;
    if constexpr (AI2C) {  // pixel (0, 0) of token r's 16 x 16 window, channel 0 (ALoader::init above)
        const int tps = p.im_nty * p.im_ntx;
        const int b = p.im_seq0 + r / tps, t = r % tps, ty = t / p.im_ntx, tx = t % p.im_ntx;
        const int gsz = p.im.grid_w * p.im.grid_h, bi = b / gsz, sq = b % gsz, p1 = sq / p.im.grid_h, p2 = sq % p.im.grid_h;
        ap = (const T*)p.A + (int64_t)bi * p.im.batch_stride + (int64_t)(p1 * p.im.patch_h + ty * 16) * p.im.row_stride + p2 * p.im.patch_w + tx * 16;
    }
    const T* wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int n = n0 + j * 16 + li;
        n = n < p.N ? n : p.N - 1;
        wp[j] = (const T*)p.W + (int64_t)n * p.ldw + g * (EVEN && !AI2C && !ASC ? 2 * EPC : EPC);
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const int nk = p.K / KC;
    // a ring of RING k-steps of operands in registers: a step's products are issued RING - 1 requests behind its loads, so that
    // the L2 round trip is paid once per RING steps instead of once per step (fc2's K = 1536 is 48 steps in bf16).
    constexpr int RING = SMALL_RING;
    u32x4 a[RING], w0[RING], w1[RING];
    if constexpr (EVEN) {
        // nk a multiple of RING (every Linear of the two ViTs: K = 192 .. 1536): no guards in the loop, so that hipcc counts the
        // loads in flight instead of draining them at every conditional.  The walk over k starts at a step that depends on the
        // COLUMN tile (and wraps): with a row pitch of 3072 B (K = 1536 in bf16) the 16 rows of a request fall on 4 of the 16 L2
        // channels at any one k -- staggered starts spread the waves of a row tile over all of them.  The summation order of an
        // output element depends on its column only, not on which rows share the call.
        const int rot = AI2C || ASC ? 0 : ((n0 >> 5) * 4) % nk;
        auto kstep = [&](int j) {
            int kk = rot + j;
            return kk >= nk ? kk - nk : kk;
        };
        auto woff = [&](int kk) { return AI2C || ASC ? kk * KC : pair_off<EPC>(kk); };
        auto aoff = [&](int kk) {
            if constexpr (AI2C) {
                const int k = kk * KC + g * EPC;  // (the lane part is not in `ap` here)
                return (k >> 8) * i2c_cs + ((k >> 4) & 15) * i2c_rs + (k & 15);
            }
            if constexpr (ASC) return AIMG ? kk * 512 : kk * KC;
            return AIMG ? (kk >> 1) * 1024 + (kk & 1) * 128 : pair_off<EPC>(kk);
        };
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            const int kk = kstep(d), kn = woff(kk);
            a[d] = *(const u32x4*)(ap + aoff(kk)), w0[d] = *(const u32x4*)(wp[0] + kn), w1[d] = *(const u32x4*)(wp[1] + kn);
        }
        for (int k = 0; k + RING < nk; k += RING) {
#pragma unroll
            for (int d = 0; d < RING; ++d) {
                Tr<T>::mma16(acc[0], w0[d], a[d]);
                Tr<T>::mma16(acc[1], w1[d], a[d]);
                const int kk = kstep(k + d + RING), kn = woff(kk);
                a[d] = *(const u32x4*)(ap + aoff(kk)), w0[d] = *(const u32x4*)(wp[0] + kn), w1[d] = *(const u32x4*)(wp[1] + kn);
            }
        }
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            Tr<T>::mma16(acc[0], w0[d], a[d]);
            Tr<T>::mma16(acc[1], w1[d], a[d]);
        }
    } else {  // any K that is a multiple of KC: k ascending in 64-byte steps, guarded ring
#pragma unroll
        for (int d = 0; d < RING; ++d) {
            const int kn = (d < nk ? d : nk - 1) * KC;
            a[d] = *(const u32x4*)(ap + kn), w0[d] = *(const u32x4*)(wp[0] + kn), w1[d] = *(const u32x4*)(wp[1] + kn);
        }
        for (int k = 0; k < nk; k += RING) {
#pragma unroll
            for (int d = 0; d < RING; ++d) {
                if (k + d < nk) {
                    Tr<T>::mma16(acc[0], w0[d], a[d]);
                    Tr<T>::mma16(acc[1], w1[d], a[d]);
                }
                if (k + d + RING < nk) {
                    const int kn = (k + d + RING) * KC;
                    a[d] = *(const u32x4*)(ap + kn), w0[d] = *(const u32x4*)(wp[0] + kn), w1[d] = *(const u32x4*)(wp[1] + kn);
                }
            }
        }
    }
    if (row0 + li >= p.M) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + 4 * g;
        if (n < p.N) epilogue<T, FLAGS>(p, row0 + li, n, acc[j]);
    }
}

// The same kernel with LayerNorm in its prologue (GemmParams::ln_w set; A = the fp32 residual rows, K = 384 or 192): the lane that
// supplies lane group g's chunks of row `li` holds, over all k-steps, exactly a quarter of that row -- K / 4 values, columns
// 8 EPC m + 2 EPC g + (0 .. 2 EPC - 1) for every step pair m; the other three quarters sit in lanes li + 16, + 32, + 48.  Row
// statistics are therefore a sum over the lane's own values and two cross-lane steps (two-pass, like misc.hip's ln_kernel), and
// the normalised row never leaves the registers: it is rounded to T straight into the MFMA operands of all k-steps.  Every wave
// of a row tile repeats the statistics of its 16 rows (N / 32 times 24 KiB out of L2 at K = 384): a small call is bound by its
// chain of launches, not by L2 bytes, and this removes two launches of the seven a block has.
template <typename T, int FLAGS, int K>
__global__ __launch_bounds__(64) void lngemm_small_kernel(const GemmParams p) {
    const int lane = threadIdx.x;
    const int li = lane & 15, g = lane >> 4;
    const int tiles_n = (p.N + 31) / 32;
    const int row0 = (blockIdx.x / tiles_n) * 16, n0 = (blockIdx.x % tiles_n) * 32;
    constexpr int EPC = Tr<T>::EPC, KC = 4 * EPC, nk = K / KC;  // (K is a template parameter: no guard, every load in one flight)
    static_assert(nk % 2 == 0 && (K / 2) % 4 == 0, "paired k order");
    int r = row0 + li;
    r = r < p.M ? r : p.M - 1;
    const float* ap = (const float*)p.A + (int64_t)r * p.lda + g * 2 * EPC;
    const T* wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int n = n0 + j * 16 + li;
        n = n < p.N ? n : p.N - 1;
        wp[j] = (const T*)p.W + (int64_t)n * p.ldw + g * 2 * EPC;
    }
    // one flight of requests: the weights of the first RING k-steps, gamma | beta on their way to LDS, the wave's 16 rows.
    // (sched_barrier: hipcc otherwise sinks each load to its first use and waits there, one round trip each.)
    constexpr int RING = nk < 12 ? nk : 12;
    constexpr int GBN = (K / 2 + 63) / 64;  // float4 pieces of gamma | beta per lane
    __shared__ __attribute__((aligned(16))) float gb[2 * K];
    u32x4 w0[RING], w1[RING];
#pragma unroll
    for (int d = 0; d < RING; ++d) w0[d] = *(const u32x4*)(wp[0] + pair_off<EPC>(d)), w1[d] = *(const u32x4*)(wp[1] + pair_off<EPC>(d));
    f32x4 gbv[GBN];
#pragma unroll
    for (int i = 0; i < GBN; ++i) {
        const int idx = lane + 64 * i;  // piece idx of gamma (idx < K / 4) or beta
        if (idx < K / 2) gbv[i] = *(const f32x4*)(idx < K / 4 ? p.ln_w + 4 * idx : p.ln_b + 4 * (idx - K / 4));
    }
    float v[nk][EPC];
#pragma unroll
    for (int s = 0; s < nk; ++s) {
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
            const f32x4 t = *(const f32x4*)(ap + pair_off<EPC>(s) + e);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[s][e + i] = t[i];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < GBN; ++i) {
        const int idx = lane + 64 * i;
        if (idx < K / 2) *(f32x4*)(&gb[4 * idx]) = gbv[i];
    }
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < nk; ++s)
#pragma unroll
        for (int e = 0; e < EPC; ++e) sum += v[s][e];
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)K;
    float q = 0.f;
#pragma unroll
    for (int s = 0; s < nk; ++s)
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const float c = v[s][e] - mean;
            q += c * c;
        }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q / (float)K + p.ln_eps);
    __syncthreads();  // (one wave: orders the LDS writes above before the reads below)
    u32x4 a[nk];
#pragma unroll
    for (int s = 0; s < nk; ++s) {
        float o[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
            const int col = pair_off<EPC>(s) + g * 2 * EPC + e;
            const f32x4 gm = *(const f32x4*)(&gb[col]), bt = *(const f32x4*)(&gb[K + col]);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[e + i] = (v[s][e + i] - mean) * rstd * gm[i] + bt[i];
        }
        if constexpr (EPC == 8) {
            a[s][0] = pack_bf16x2(o[0], o[1]), a[s][1] = pack_bf16x2(o[2], o[3]);
            a[s][2] = pack_bf16x2(o[4], o[5]), a[s][3] = pack_bf16x2(o[6], o[7]);
        } else {
            f32x4 t;
            t[0] = o[0], t[1] = o[1], t[2] = o[2], t[3] = o[3];
            a[s] = __builtin_bit_cast(u32x4, t);
        }
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < nk; ++s) {
        Tr<T>::mma16(acc[0], w0[s % RING], a[s]);
        Tr<T>::mma16(acc[1], w1[s % RING], a[s]);
        if (s + RING < nk) w0[s % RING] = *(const u32x4*)(wp[0] + pair_off<EPC>(s + RING)), w1[s % RING] = *(const u32x4*)(wp[1] + pair_off<EPC>(s + RING));
    }
    if (row0 + li >= p.M) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + j * 16 + 4 * g;
        if (n < p.N) epilogue<T, FLAGS>(p, row0 + li, n, acc[j]);
    }
}

template <typename T, int ALOAD, int FLAGS>
int launch(const GemmParams& p, hipStream_t st) {
    if constexpr (ALOAD == ALOAD_IM2COL) {
        // (image strides whose largest in-patch offset, 2 channel planes + 15 pixel rows + 15, fits the kernel's 32-bit arithmetic)
        if (p.M <= SMALL_M && p.K == 768 && p.im.chan_stride >= 0 && p.im.row_stride >= 0 &&
            2 * (int64_t)p.im.chan_stride + 15 * (int64_t)p.im.row_stride + 16 < (1ll << 31) && !hipt_generic_only()) {
            hipLaunchKernelGGL((gemm_small_kernel<T, FLAGS, true, false, true>), dim3(((p.M + 15) / 16) * ((p.N + 31) / 32)), dim3(64), 0, st, p);
            HIPT_CHECK_LAUNCH();
            return HIPT_OK;
        }
    }
    if constexpr (ALOAD == ALOAD_PLAIN) {
        const dim3 sgrid(((p.M + 15) / 16) * ((p.N + 31) / 32));
        if (p.ln_w) {  // (checked by hipt_gemm_launch: small M, K = 384 or 192, the two plain epilogues)
            if constexpr (FLAGS == 0 || FLAGS == HIPT_EPI_GELU) {
                if (p.K == 384)
                    hipLaunchKernelGGL((lngemm_small_kernel<T, FLAGS, 384>), sgrid, dim3(64), 0, st, p);
                else
                    hipLaunchKernelGGL((lngemm_small_kernel<T, FLAGS, 192>), sgrid, dim3(64), 0, st, p);
                HIPT_CHECK_LAUNCH();
                return HIPT_OK;
            }
        }
        if (p.a_row_step > 0) {  // (checked by hipt_gemm_launch: bf16, small M, K = 384, plain epilogue)
            if constexpr (FLAGS == 0 && sizeof(T) == 2) {
                if (p.asc) hipLaunchKernelGGL((gemm_small_kernel<T, 0, true, true, false, true>), sgrid, dim3(64), 0, st, p);
                else hipLaunchKernelGGL((gemm_small_kernel<T, 0, true, true>), sgrid, dim3(64), 0, st, p);
                HIPT_CHECK_LAUNCH();
                return HIPT_OK;
            }
        }
        if ((p.M <= SMALL_M || p.small_any) && p.K % (4 * Tr<T>::EPC) == 0) {
            if (p.asc && p.K % (SMALL_RING * 4 * Tr<T>::EPC) == 0) {
                if constexpr (FLAGS == 0) hipLaunchKernelGGL((gemm_small_kernel<T, 0, true, false, false, true>), sgrid, dim3(64), 0, st, p);
                else hipLaunchKernelGGL((gemm_small_kernel<T, FLAGS, false>), sgrid, dim3(64), 0, st, p);  // (the guarded ring walks k in ascending order too)
            } else if (p.K % (SMALL_RING * 4 * Tr<T>::EPC) == 0)
                hipLaunchKernelGGL((gemm_small_kernel<T, FLAGS, true>), sgrid, dim3(64), 0, st, p);
            else
                hipLaunchKernelGGL((gemm_small_kernel<T, FLAGS, false>), sgrid, dim3(64), 0, st, p);
            HIPT_CHECK_LAUNCH();
            return HIPT_OK;
        }
    }
    const int tiles = ((p.M + p.rpt - 1) / p.rpt) * ((p.N + BN - 1) / BN);
    static DevOnce once;  // > 64 KiB dynamic LDS needs the opt-in once per kernel and device
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)gemm_kernel<T, ALOAD, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                GEMM_LDS) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(gemm) failed");
            return HIPT_E_LAUNCH;
        }
        once.done[dev] = true;
    }
    hipLaunchKernelGGL((gemm_kernel<T, ALOAD, FLAGS>), dim3(tiles), dim3(512), GEMM_LDS, st, p);
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

// (M: the sequence-tiled kernel's DMA could address the rows too -- built and measured -- but every one of its N / 128 column tiles then re-reads
//  each row as 48 separate 16-byte pieces of as many cache lines: 34 us per block at 2 048 [CLS] rows against 5 + 15 for a gather
//  launch and the dense GEMM.  Up to SMALL_M rows the launch saved is worth more than the scattered reads cost.)
bool hipt_gemm_arows_supported(int M, int K, int dtype, int aload, int flags) {
    return dtype == HIPT_BF16 && M <= SMALL_M && K == 384 && aload == ALOAD_PLAIN && flags == 0;
}

bool hipt_gemm_ln_supported(int M, int K, int aload, int flags, bool any_m) {
    return (M <= SMALL_M || any_m) && (K == 384 || K == 192) && aload == ALOAD_PLAIN && (flags == 0 || flags == HIPT_EPI_GELU);
}

int hipt_gemm_launch(const GemmParams& p_in, int dtype, int aload, int flags, hipStream_t st) {
    GemmParams p = p_in;
    const int kb = dtype == HIPT_F32 ? 32 : 64;
    HIPT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    HIPT_CHECK_ARG(p.K % kb == 0, "gemm: K=%d must be a multiple of %d", p.K, kb);
    HIPT_CHECK_ARG(p.N % 4 == 0, "gemm: N=%d must be a multiple of 4", p.N);
    HIPT_CHECK_ARG(p.ldc % 4 == 0, "gemm: ldc=%lld must be a multiple of 4", (long long)p.ldc);
    if (p.rpt <= 0 || p.rpt > TROWS) p.rpt = TROWS;  // no sequence structure: plain 272-row tiles
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
    if (p.a_row_step > 0) {
        HIPT_CHECK_ARG(hipt_gemm_arows_supported(p.M, p.K, dtype, aload, flags) && !p.ln_w,
                       "gemm: rows gathered from an activation image take bf16, M <= %d, K = 384, plain loader and epilogue", SMALL_M);
    }
    if (p.ln_w)
        HIPT_CHECK_ARG(hipt_gemm_ln_supported(p.M, p.K, aload, flags, p.small_any != 0) && p.ln_b && (p.lda * 4) % 16 == 0 && ((uintptr_t)p.ln_w % 16) == 0 &&
                           ((uintptr_t)p.ln_b % 16) == 0,
                       "gemm: the LayerNorm prologue takes M <= %d fp32 rows of K = 384 or 192, plain loader, no or GELU epilogue", SMALL_M);
    if (dtype == HIPT_F32) return dispatch<float>(p, aload, flags, st);
    if (dtype == HIPT_BF16) return dispatch<bf16_t>(p, aload, flags, st);
    hipt_set_error("gemm: bad dtype %d", dtype);
    return HIPT_E_BADARG;
}
