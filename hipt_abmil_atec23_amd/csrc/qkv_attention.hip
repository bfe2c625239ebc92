// QKV projection + softmax attention of one LayerNorm-chained ViT-256 block in ONE kernel (Attention.forward,
// HIPT_4K/vision_transformer.py:119-128, lines 121-128: qkv Linear, q k^T * scale, softmax, P v): the q | k | v tensor
// ([M, 1152] bf16, 1.2 GB per 2048 patches, written by the QKV GEMM and read back by the attention kernel) never exists.
//
// Shape: D = 384, 6 heads of 64, 257 tokens per sequence (one 256 x 256 patch: 256 tokens + [CLS]); bf16 operands, fp32
// accumulation, fp32 softmax.  Input xn = LayerNorm-1(x) as a bf16 ACTIVATION IMAGE (kernels.h), written by the previous
// block's fused MLP; output = the attention output (before proj) as one, read by the proj GEMM.
//
// One persistent 8-wave workgroup per CU walks patches; per patch it walks the 6 heads:
//   GEMM phase   wave w owns the patch's tokens 1 + 32 w .. 32 w + 32 as ONE 32-column B operand of v_mfma_f32_32x32x16_bf16
//                (24 k-steps = 96 registers, loaded from the image; re-loaded per head because the scores need the registers);
//                the head's [192, 384] weight slice streams through a 3 x 24 KiB LDS-DMA ring as six units of 32 output
//                columns (K K V V Q Q), every unit 24 A fragments of 1 KiB in operand order (image made once by
//                hipt_qkv_attn_pack_launch: a DMA piece and a fragment read are 1 KiB of consecutive bytes).  D = W X^T lands
//                with the token on the lane and the output column in the registers:
//                  K^T tiles -> packed and written as the A-operand fragments of the score product (1 KiB per (key tile, k-step));
//                  V^T tiles -> written row-major [key][32 dims] (64-byte rows) for the transposing LDS read;
//                  Q^T tiles -> converted in place into the B operand of the score product (accumulator-as-operand).
//   attention    S^T[key][query] = K Q^T per 32-key tile (9 tiles: 8 of patch tokens + one holding the [CLS] key), softmax over
//                the registers + one cross-half exchange, P^T packed in place as the B operand of O^T = V^T P^T.  No barrier
//                inside the phase: the two waves of a SIMD drift apart, one's exponentials run under the other's MFMAs.
//   [CLS]        257 = 8 x 32 + 1.  The [CLS] row's q | k | v come from a side GEMM over the nseq [CLS] rows (capi.hip), staged
//                per patch into LDS by DMA.  Its key / value are row 256 of the K / V images; its QUERY is spread over the
//                waves by key range: wave w does the [CLS] query against its own 32 keys (8 + 6 MFMAs with one live column),
//                the partial (max, sum, o[64]) goes through LDS and one wave merges the eight.
// LDS: ring 72 KiB | K image 36 KiB | V image 36 KiB | bias 4.5 KiB | [CLS] partials | [CLS] q k v rows = 158.75 KiB.
// HBM per patch: xn read (197 KB; the five re-reads per patch are L2 / MALL hits) + output written (197 KB).
#include "common.h"
#include "kernels.h"
#include "pipe_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int D = 384, NTOK = 257, HEADS = 6;
constexpr int UNIT = 24576, NUNIT = 36;                  // one unit = 32 output columns x 384 k = 24 fragments of 1 KiB
constexpr int OFF_K = 3 * UNIT;                          // K image: [9 key tiles][4 (d tile, k-step)][1 KiB]
constexpr int VSUB = 288 * 64;                           // V image: two [288 keys][32 dims] sub-images
constexpr int OFF_V = OFF_K + 9 * 4096;
constexpr int OFF_BIAS = OFF_V + 2 * VSUB;               // [36 units][2 lane halves][16] floats in accumulator order
constexpr int CLSP_W = 68 * 4;                           // one wave's [CLS] partial: m, l, -, -, o[64]
constexpr int OFF_CLSP = OFF_BIAS + NUNIT * 32 * 4;      // [2 (head parity)][8 waves]
constexpr int CLSROW = 3072;                             // q | k | v of one [CLS] row (2 304 B), DMA'd as three 1 KiB pieces
constexpr int OFF_CLSROW = OFF_CLSP + 2 * 8 * CLSP_W;    // [2 (patch parity)]
constexpr int LDS_BYTES = OFF_CLSROW + 2 * CLSROW;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

// row of qkv_w (= output column of the QKV Linear) that A-operand row c of unit U holds: units of a head are K K V V Q Q
__host__ __device__ __forceinline__ int unit_row(int U, int c) {
    const int h = U / 6, u = U % 6;
    const int base = u < 2 ? D + 64 * h + 32 * u : (u < 4 ? 2 * D + 64 * h + 32 * (u - 2) : 64 * h + 32 * (u - 4));
    return base + c;
}

// one thread per 16-byte chunk of the image: unit U, fragment s, lane (r, hh) <- W[unit_row(U, r)][16 s + 8 hh ..+7]
__global__ void qkv_attn_pack_kernel(const bf16_t* __restrict__ W, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NUNIT * 24 * 64) return;
    const int U = i / (24 * 64), s = (i >> 6) % 24, lane = i & 63;
    out[i] = *(const u32x4*)(W + (int64_t)unit_row(U, lane & 31) * D + 16 * s + 8 * (lane >> 5));
}

#define DSRTR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define DSW128(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define DSW64(addr, val, off) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define DSW32(addr, val, off) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(val), "n"(off) : "memory")
#define GLD128(dst, ptr, off) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(ptr), "n"(off))

__device__ __forceinline__ u32x2 lds_ld64w(uint32_t a) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}

__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// registers 8 s .. 8 s + 7 of an accumulator tile as one bf16 operand fragment (accumulator-as-operand: k slot (half hh,
// element j) <-> tile row 16 s + 8 (j >> 2) + 4 hh + (j & 3); the other operand is read / written in that same order)
template <int S>
__device__ __forceinline__ u32x4 pack8(const f32x16& a) {
    u32x4 o;
    o[0] = pack_bf16x2(a[8 * S + 0], a[8 * S + 1]);
    o[1] = pack_bf16x2(a[8 * S + 2], a[8 * S + 3]);
    o[2] = pack_bf16x2(a[8 * S + 4], a[8 * S + 5]);
    o[3] = pack_bf16x2(a[8 * S + 6], a[8 * S + 7]);
    return o;
}

struct QkvAttnParams {
    const char* xn;        // bf16 activation image [M, 384]: LayerNorm-1(x)
    const char* wpk;       // the weight image (hipt_qkv_attn_pack_launch)
    const float* bias;     // qkv_b [1152]
    const char* qkv_cls;   // bf16 [nseq][1152] (+ 1 KiB of slack): q | k | v of the [CLS] rows
    char* out;             // bf16 activation image [M, 384]: attention output
    int nseq;
    float sl2e;            // scale * log2(e)
    unsigned out_bytes;
};

__global__ __launch_bounds__(512, 2) void qkv_attn_kernel(const QkvAttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;

    // ---- one-time LDS contents (no DMA in flight yet: plain stores) ----
    {
        float* bias_s = (float*)(smem + OFF_BIAS);
        for (int i = tid; i < NUNIT * 32; i += 512) {
            const int U = i >> 5, hb = (i >> 4) & 1, ii = i & 15;
            bias_s[i] = p.bias[unit_row(U, (ii & 3) + 8 * (ii >> 2) + 4 * hb)];
        }
        uint32_t* kz = (uint32_t*)(smem + OFF_K + 8 * 4096);  // key tile 8: row 0 = the [CLS] key (written per head), rows 1.. stay zero
        for (int i = tid; i < 1024; i += 512) kz[i] = 0u;
        for (int t = 0; t < 2; ++t) {                          // V rows 256 .. 287: row 256 = the [CLS] value, the rest stay zero
            uint32_t* vz = (uint32_t*)(smem + OFF_V + t * VSUB + 256 * 64);
            for (int i = tid; i < 512; i += 512) vz[i] = 0u;
        }
    }
    __syncthreads();

    const uint32_t lbase = lds_addr(smem);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, (int)p.out_bytes, 0x00020000);

    // ---- weight stream: unit n of this workgroup = image unit n % 36 into ring slot n % 3; wave w moves pieces w, w + 8, w + 16 ----
    int iU = 0, islot = 0;
    auto issue_unit = [&]() __attribute__((always_inline)) {
        const char* src = p.wpk + (size_t)iU * UNIT + lane * 16;
        char* dst = smem + islot * UNIT;
#pragma unroll
        for (int j = 0; j < 3; ++j) glds16(src + (w + 8 * j) * 1024, dst + (w + 8 * j) * 1024);
        iU = iU + 1 == NUNIT ? 0 : iU + 1;
        islot = islot + 1 == 3 ? 0 : islot + 1;
    };
    auto cls_dma = [&](int b, int par) __attribute__((always_inline)) {  // (wave 0) the [CLS] row of patch b -> LDS
        const char* src = p.qkv_cls + (size_t)b * (3 * D * 2) + lane * 16;
#pragma unroll
        for (int j = 0; j < 3; ++j) glds16(src + j * 1024, smem + OFF_CLSROW + par * CLSROW + j * 1024);
    };

    // this wave's tokens as the B operand: 24 k-steps, lane (r, hh) holds row R, 16-byte chunk 2 s + hh of the image
    u32x4 xop[24];
    auto load_xop = [&](int b) __attribute__((always_inline)) {
        const int64_t R = (int64_t)b * NTOK + 1 + 32 * w + r;
        const char* x0 = p.xn + (R >> 4) * 12288 + (R & 15) * 16 + hh * 256;
        const char* x1 = x0 + 4096;
        const char* x2 = x0 + 8192;
        sfor<0, 24>([&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            u32x4& d = xop[s];
            const char* a0 = x0;  // (plain uses: an asm operand inside `if constexpr` alone does not capture the variable)
            const char* a1 = x1;
            const char* a2 = x2;
            if constexpr (s < 8) GLD128(d, a0, s * 512);
            else if constexpr (s < 16) GLD128(d, a1, (s - 8) * 512);
            else GLD128(d, a2, (s - 16) * 512);
        });
    };
    // (the asm loads land asynchronously: nothing may touch xop between them and this statement, which is the counted wait
    //  AND the point from which the compiler may use the registers)
#define XOP_FENCE(N)                                                                                                              \
    asm volatile("s_waitcnt vmcnt(" #N ") ; XOP_FENCE"                                                                                       \
                 : "+v"(xop[0]), "+v"(xop[1]), "+v"(xop[2]), "+v"(xop[3]), "+v"(xop[4]), "+v"(xop[5]), "+v"(xop[6]), "+v"(xop[7]), \
                   "+v"(xop[8]), "+v"(xop[9]), "+v"(xop[10]), "+v"(xop[11]), "+v"(xop[12]), "+v"(xop[13]), "+v"(xop[14]),           \
                   "+v"(xop[15]), "+v"(xop[16]), "+v"(xop[17]), "+v"(xop[18]), "+v"(xop[19]), "+v"(xop[20]), "+v"(xop[21]),         \
                   "+v"(xop[22]), "+v"(xop[23])::"memory")

    // ---- merge of the eight [CLS]-query partials of one (patch, head): one wave, lane = output dimension ----
    auto merge_cls = [&](int b, int h, int pq) __attribute__((always_inline)) {
        int ln = lane;  // (opaque copy: see the head loop)
        asm volatile("" : "+v"(ln));
        const uint32_t base = lbase + OFF_CLSP + pq * 8 * CLSP_W, obase = base + 16 + ln * 4;
        float mk[8], lk[8], ok[8];
        sfor<0, 8>([&](auto K_) __attribute__((always_inline)) {
            constexpr int k = decltype(K_)::value;
            float &m_ = mk[k], &l_ = lk[k], &o_ = ok[k];
            const uint32_t ba = base, oa = obase;
            asm volatile("ds_read_b32 %0, %3 offset:%5\n\tds_read_b32 %1, %3 offset:%6\n\tds_read_b32 %2, %4 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(m_), "=&v"(l_), "=&v"(o_)
                         : "v"(ba), "v"(oa), "n"(k * CLSP_W), "n"(k * CLSP_W + 4));
        });
        float mx = mk[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) mx = fmaxf(mx, mk[k]);
        float L = 0.f, o = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float f = __builtin_amdgcn_exp2f((mk[k] - mx) * p.sl2e);
            L += f * lk[k];
            o += f * ok[k];
        }
        const int64_t R = (int64_t)b * NTOK;  // the [CLS] row; column 64 h + lane = chunk 8 h + (lane >> 3), element lane & 7
        const int kc = 8 * h + (ln >> 3);
        bf16_t* dst = (bf16_t*)(p.out + (R >> 4) * 12288 + (kc >> 2) * 1024 + (kc & 3) * 256 + (R & 15) * 16) + (ln & 7);
        *dst = (bf16_t)(o / L);
    };

    int pq = 0;                       // parity of the (patch, head) counter: which [CLS] partial buffer
    int prev_b = -1, prev_h = 0;      // (patch, head) whose partials wait for their merge
    int par = 0;                      // parity of the patch counter: which [CLS] row buffer
    const int b0 = blockIdx.x, bstep = gridDim.x;
    if (w == 0) cls_dma(b0, 0);
    issue_unit();
    issue_unit();
    load_xop(b0);

    for (int b = b0; b < p.nseq; b += bstep, par ^= 1) {
        for (int h = 0; h < HEADS; ++h, pq ^= 1) {
            // Per-lane addresses are re-derived per head from an opaque copy of the lane id: left loop-invariant, hipcc hoists a
            // dozen of them out of the head loop, spills them across the attention phase and reloads them inside the ring phases --
            // and every scratch reload waits vmcnt(0), i.e. for the weight stream.
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int r = ln & 31, hh = ln >> 5;
            const uint32_t fa = lbase + ln * 16;  // this lane's 16 bytes of a 1 KiB fragment
            // transposing V read: lane 4 q + pp of a 16-lane group supplies row (key) q, columns 4 pp .. of the group's 4 x 16 block
            const uint32_t va = lbase + OFF_V + (4 * hh + ((ln & 15) >> 2)) * 64 + (16 * ((ln >> 4) & 1) + 4 * (ln & 3)) * 2;
            const int64_t Rl = (int64_t)b * NTOK + 1 + 32 * w + r;                          // this lane's token row (as a query)
            const int orow = (int)((Rl >> 4) * 12288 + (Rl & 15) * 16) + 8 * hh;            // its bytes in the output image
            u32x4 qop[2][2];
            // ================= GEMM phase: six ring units =================
            sfor<0, 6>([&](auto U_) __attribute__((always_inline)) {
                constexpr int u = decltype(U_)::value;
                // my pieces of this unit have landed (all but the three youngest vector-memory operations are complete: those
                // are at most the next unit's pieces) ... everyone's; and everyone is done with the unit before it
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const uint32_t sa = fa + (islot == 2 ? 0 : islot + 1) * UNIT;  // slot of the unit to consume = (issue slot + 1) % 3
                issue_unit();
                if constexpr (u == 0) {
                    // behind the first barrier of a head every wave has left the previous head's attention: its [CLS] partials are
                    // complete (merge them), and the other [CLS] row buffer is free (fetch the next patch's row)
                    if (prev_b >= 0 && w == (prev_h & 7)) merge_cls(prev_b, prev_h, pq ^ 1);
                    if (h == 0 && w == 0 && b + bstep < p.nseq) cls_dma(b + bstep, par ^ 1);
                    XOP_FENCE(3);
                }
                f32x16 acc;
                {
                    const uint32_t ba = lbase + OFF_BIAS + ((h * 6 + u) * 2 + hh) * 64;
                    f32x4 b0v, b1v, b2v, b3v;
                    DSR128X4_WAIT(b0v, b1v, b2v, b3v, ba, 0, 16, 32, 48);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i] = b0v[i];
                        acc[4 + i] = b1v[i];
                        acc[8 + i] = b2v[i];
                        acc[12 + i] = b3v[i];
                    }
                }
                u32x4 wf[4];
                {
                    u32x4 &w0 = wf[0], &w1 = wf[1], &w2 = wf[2];
                    DSR128(w0, sa, 0);
                    DSR128(w1, sa, 1024);
                    DSR128(w2, sa, 2048);
                }
                sfor<0, 24>([&](auto S_) __attribute__((always_inline)) {
                    constexpr int s = decltype(S_)::value;
                    if constexpr (s + 3 < 24) {
                        u32x4& wn = wf[(s + 3) & 3];
                        DSR128(wn, sa, (s + 3) * 1024);
                        LGKM(3);
                    } else {
                        LGKM(23 - s);
                    }
                    acc = mfma32(wf[s & 3], xop[s], acc);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (u < 2) {          // K^T tile u: the two A-operand fragments (k-steps) of the score product
                    const u32x4 k0 = pack8<0>(acc), k1 = pack8<1>(acc);
                    const uint32_t ka = fa + OFF_K + w * 4096;
                    DSW128(ka, k0, (u * 2 + 0) * 1024);
                    DSW128(ka, k1, (u * 2 + 1) * 1024);
                } else if constexpr (u < 4) {   // V^T tile: row-major [key][32 dims], dims 8 q + 4 hh ..+3 from registers 4 q ..+3
                    const uint32_t vw = lbase + OFF_V + (u - 2) * VSUB + (32 * w + r) * 64 + 8 * hh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        u32x2 o;
                        o[0] = pack_bf16x2(acc[4 * q], acc[4 * q + 1]);
                        o[1] = pack_bf16x2(acc[4 * q + 2], acc[4 * q + 3]);
                        if (q == 0) DSW64(vw, o, 0);
                        else if (q == 1) DSW64(vw, o, 16);
                        else if (q == 2) DSW64(vw, o, 32);
                        else DSW64(vw, o, 48);
                    }
                } else {                        // Q^T tile: stays in registers as the score product's B operand
                    qop[u - 4][0] = pack8<0>(acc);
                    qop[u - 4][1] = pack8<1>(acc);
                }
            });
            // ---- the [CLS] token's key and value of this head: K image tile 8 row 0, V image row 256 ----
            const uint32_t crow = lbase + OFF_CLSROW + par * CLSROW;
            if (w == 0 && r == 0) {  // lanes 0 and 32: the two lane halves of row 0
#pragma unroll
                for (int f = 0; f < 4; ++f) {  // fragment (d tile f >> 1, k-step f & 1): dims 16 f + 4 hh ..+3 and 16 f + 8 + 4 hh ..+3
                    const uint32_t ksrc = crow + (D + 64 * h + 16 * f + 4 * hh) * 2;
                    const u32x2 lo = lds_ld64w(ksrc), hi = lds_ld64w(ksrc + 16);
                    const u32x4 kv = {lo[0], lo[1], hi[0], hi[1]};
                    const uint32_t ka = fa + OFF_K + 8 * 4096 + f * 1024;
                    DSW128(ka, kv, 0);
                }
            }
            if (w == 1 && lane < 16) {  // 2 sub-images x 8 pieces of 8 bytes
                const int t = lane >> 3, part = lane & 7;
                const u32x2 vv = lds_ld64w(crow + (2 * D + 64 * h + 32 * t + 4 * part) * 2);
                const uint32_t vd = lbase + OFF_V + t * VSUB + 256 * 64 + part * 8;
                DSW64(vd, vv, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // K / V images of this head are complete

            // ================= the [CLS] query against this wave's keys (wave 0: + the [CLS] key) =================
            // (first, while few registers are live: the partial is merged behind the next head's first barrier)
            {
                u32x4 qc[4];  // B operand with one live column (query 0 = lanes 0 and 32): fragment f = (d tile, k-step)
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const uint32_t qsrc = crow + (64 * h + 16 * f + 4 * hh) * 2;
                    const u32x2 lo = lds_ld64w(qsrc), hi = lds_ld64w(qsrc + 16);
                    qc[f] = r == 0 ? u32x4{lo[0], lo[1], hi[0], hi[1]} : u32x4{0u, 0u, 0u, 0u};
                }
                f32x16 Sc[2];
                const uint32_t ko = fa + OFF_K + w * 4096, k8 = fa + OFF_K + 8 * 4096;
                {
                    u32x4 a0, a1, a2, a3;
                    DSR128X4_WAIT(a0, a1, a2, a3, ko, 0, 1024, 2048, 3072);
                    Sc[0] = mfma32(a0, qc[0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                    Sc[0] = mfma32(a1, qc[1], Sc[0]);
                    Sc[0] = mfma32(a2, qc[2], Sc[0]);
                    Sc[0] = mfma32(a3, qc[3], Sc[0]);
                    DSR128X4_WAIT(a0, a1, a2, a3, k8, 0, 1024, 2048, 3072);
                    Sc[1] = mfma32(a0, qc[0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                    Sc[1] = mfma32(a1, qc[1], Sc[1]);
                    Sc[1] = mfma32(a2, qc[2], Sc[1]);
                    Sc[1] = mfma32(a3, qc[3], Sc[1]);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (w != 0 || i != 0 || hh != 0) Sc[1][i] = -INFINITY;  // the [CLS] key belongs to wave 0's share
                float mc = Sc[1][0];
#pragma unroll
                for (int i = 0; i < 16; ++i) mc = fmaxf(mc, Sc[0][i]);
                mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
                const float mcs = -mc * p.sl2e;
                float lc = 0.f;
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(Sc[k][i], p.sl2e, mcs));
                        lc += e;
                        Sc[k][i] = e;
                    }
                lc += __shfl_xor(lc, 32, 64);
                f32x16 Oc[2];
                const u32x4 pc0 = pack8<0>(Sc[0]), pc1 = pack8<1>(Sc[0]), pc8 = pack8<0>(Sc[1]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const uint32_t vo = va + t * VSUB + 32 * w * 64, v8 = va + t * VSUB + 256 * 64;
                    u32x2 a0, a1, b0, b1, c0, c1;
                    asm volatile("ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %6 offset:512\n\tds_read_b64_tr_b16 %2, %6 offset:1024\n\t"
                                 "ds_read_b64_tr_b16 %3, %6 offset:1536\n\tds_read_b64_tr_b16 %4, %7\n\tds_read_b64_tr_b16 %5, %7 offset:512\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1), "=&v"(c0), "=&v"(c1)
                                 : "v"(vo), "v"(v8));
                    Oc[t] = mfma32(u32x4{a0[0], a0[1], a1[0], a1[1]}, pc0, f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                    Oc[t] = mfma32(u32x4{b0[0], b0[1], b1[0], b1[1]}, pc1, Oc[t]);
                    Oc[t] = mfma32(u32x4{c0[0], c0[1], c1[0], c1[1]}, pc8, Oc[t]);
                }
                if (r == 0) {  // query 0: lane 0 holds dims 8 q + 0..3, lane 32 dims 8 q + 4..7 of each d tile
                    const uint32_t pa = lbase + OFF_CLSP + (pq * 8 + w) * CLSP_W;
                    if (hh == 0) {
                        DSW32(pa, mc, 0);
                        DSW32(pa, lc, 4);
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v = {Oc[t][4 * q], Oc[t][4 * q + 1], Oc[t][4 * q + 2], Oc[t][4 * q + 3]};
                            const uint32_t da = pa + 16 + (32 * t + 8 * q + 4 * hh) * 4;
                            DSW128(da, v, 0);
                        }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            // ================= attention of this wave's 32 queries =================
            f32x16 S[9];
            {
                u32x4 kf[4];
                const uint32_t kb = fa + OFF_K;  // (ds offsets are 16-bit: the image base travels in the address register)
                {
                    u32x4 &k0 = kf[0], &k1 = kf[1], &k2 = kf[2];
                    DSR128(k0, kb, 0);
                    DSR128(k1, kb, 1024);
                    DSR128(k2, kb, 2048);
                }
                sfor<0, 36>([&](auto I_) __attribute__((always_inline)) {
                    constexpr int i = decltype(I_)::value, kt = i >> 2, f = i & 3;
                    if constexpr (i + 3 < 36) {
                        u32x4& kn = kf[(i + 3) & 3];
                        const uint32_t ka2 = kb;
                        DSR128(kn, ka2, (i + 3) * 1024);
                        LGKM(3);
                    } else {
                        LGKM(35 - i);
                    }
                    if constexpr (f == 0) S[kt] = mfma32(kf[i & 3], qop[0][0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                    else S[kt] = mfma32(kf[i & 3], qop[f >> 1][f & 1], S[kt]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // tile 8 holds one key (register 0 of lane half 0): the rest is padding
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i != 0 || hh != 0) S[8][i] = -INFINITY;
            float m = fmaxf(S[0][0], S[0][1]);
#pragma unroll
            for (int kt = 0; kt < 9; ++kt)
#pragma unroll
                for (int i = 0; i < 16; i += 2) m = __builtin_fmaxf(__builtin_fmaxf(m, S[kt][i]), S[kt][i + 1]);  // v_max3_f32
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float ms = -m * p.sl2e;
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 9; ++kt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][i], p.sl2e, ms));
                    l += e;
                    S[kt][i] = e;
                }
            l += __shfl_xor(l, 32, 64);
            const float inv = 1.0f / l;
            // O^T = V^T P^T: 17 (key tile, k-step) pairs (tile 8: its first k-step holds the one key) x 2 d tiles
            f32x16 O[2];
            {
                u32x2 vf[4][2];
                auto rdv = [&](auto J_) __attribute__((always_inline)) {
                    constexpr int j = decltype(J_)::value, pi = j >> 1, t = j & 1, kt = pi >> 1, s = pi & 1;
                    constexpr int off = t * VSUB + (32 * kt + 16 * s) * 64;
                    u32x2 &lo = vf[j & 3][0], &hi = vf[j & 3][1];
                    const uint32_t a = va;
                    DSRTR(lo, a, off);
                    DSRTR(hi, a, off + 8 * 64);
                };
                rdv(std::integral_constant<int, 0>{});
                rdv(std::integral_constant<int, 1>{});
                u32x4 pop;
                sfor<0, 34>([&](auto J_) __attribute__((always_inline)) {
                    constexpr int j = decltype(J_)::value, pi = j >> 1, t = j & 1, kt = pi >> 1, s = pi & 1;
                    if constexpr (j + 2 < 34) {
                        rdv(std::integral_constant<int, j + 2>{});
                        LGKM(4);
                    } else {
                        LGKM((33 - j) * 2);
                    }
                    if constexpr (t == 0) pop = s == 0 ? pack8<0>(S[kt]) : pack8<1>(S[kt]);
                    const u32x4 vfrag = {vf[j & 3][0][0], vf[j & 3][0][1], vf[j & 3][1][0], vf[j & 3][1][1]};
                    if constexpr (pi == 0) O[t] = mfma32(vfrag, pop, f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f});
                    else O[t] = mfma32(vfrag, pop, O[t]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // the registers of the scores are free: request the operands of the next head (same rows) or the next patch
            // (unconditionally -- past the last patch the current one again: a conditional load would keep the OLD operands alive
            //  through the whole attention phase on the not-taken path, 96 registers the scores need)
            {
                const int nb = h + 1 < HEADS ? b : b + bstep;
                load_xop(nb < p.nseq ? nb : b);
            }
            // store: registers 4 q ..+3 of tile t = dims 32 t + 8 q + 4 hh ..+3 = chunk 8 h + 4 t + q, bytes 8 hh ..+7
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    u32x2 o;
                    o[0] = pack_bf16x2(O[t][4 * q] * inv, O[t][4 * q + 1] * inv);
                    o[1] = pack_bf16x2(O[t][4 * q + 2] * inv, O[t][4 * q + 3] * inv);
                    __builtin_amdgcn_raw_buffer_store_b64(o, orsrc, orow + h * 2048 + t * 1024 + q * 256, 0, 0);
                }

            prev_b = b;
            prev_h = h;
        }
    }
    // ---- the last (patch, head)'s [CLS] partials ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (prev_b >= 0 && w == (prev_h & 7)) merge_cls(prev_b, prev_h, pq ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (continuous stream: pieces of units nobody consumes)
}

}  // namespace

bool hipt_qkv_attn_supported(int dtype, int D_, int heads, int ntok) { return dtype == HIPT_BF16 && D_ == D && heads == HEADS && ntok == NTOK; }

size_t hipt_qkv_attn_packed_bytes() { return (size_t)NUNIT * UNIT; }

int hipt_qkv_attn_pack_launch(const void* qkv_w, void* packed, hipStream_t st) {
    const int chunks = NUNIT * 24 * 64;
    hipLaunchKernelGGL(qkv_attn_pack_kernel, dim3((chunks + 255) / 256), dim3(256), 0, st, (const bf16_t*)qkv_w, (u32x4*)packed);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}

int hipt_qkv_attn_launch(const void* xn_img, const void* wpk, const float* qkv_b, const void* qkv_cls, void* out_img, int nseq, float scale, hipStream_t st) {
    HIPT_CHECK_ARG(xn_img && wpk && qkv_b && qkv_cls && out_img && nseq > 0, "qkv_attention: null / empty argument");
    HIPT_CHECK_ARG(((int64_t)nseq * NTOK) % 16 == 0, "qkv_attention: activation images need whole 16-row fragments (nseq * 257 %% 16 == 0)");
    HIPT_CHECK_ARG((int64_t)nseq * NTOK * D * 2 < ((int64_t)1 << 32) - 65536, "qkv_attention: output image beyond 4 GiB");
    HIPT_CHECK_ARG(((uintptr_t)xn_img % 16) == 0 && ((uintptr_t)wpk % 16) == 0 && ((uintptr_t)qkv_cls % 16) == 0 && ((uintptr_t)out_img % 16) == 0,
                   "qkv_attention: 16-byte alignment required");
    static DevOnce once;
    HIPT_CUR_DEVICE(dev);
    if (!once.done[dev]) {
        if (hipFuncSetAttribute((const void*)qkv_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
            hipt_set_error("hipFuncSetAttribute(qkv_attention) failed");
            return HIPT_E_LAUNCH;
        }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            hipt_set_error("qkv_attention: cannot query the device");
            return HIPT_E_LAUNCH;
        }
        once.ncu[dev] = prop.multiProcessorCount;
        once.done[dev] = true;
    }
    QkvAttnParams p;
    p.xn = (const char*)xn_img;
    p.wpk = (const char*)wpk;
    p.bias = qkv_b;
    p.qkv_cls = (const char*)qkv_cls;
    p.out = (char*)out_img;
    p.nseq = nseq;
    p.sl2e = scale * 1.4426950408889634f;
    p.out_bytes = (unsigned)((int64_t)nseq * NTOK * D * 2);
    const int grid = nseq < once.ncu[dev] ? nseq : once.ncu[dev];
    hipLaunchKernelGGL(qkv_attn_kernel, dim3(grid), dim3(512), LDS_BYTES, st, p);
    HIPT_CHECK_LAUNCH();
    return HIPT_OK;
}
