// extern "C" entry points of libhipt_abmil.so (include/hipt_abmil.h): argument validation, scratch
// carving and the launch sequences.  Host code only: nothing here synchronises or allocates, so a
// caller may capture any call into a hipGraph.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

void hipt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {

inline size_t al256(size_t n) { return (n + 255) & ~(size_t)255; }
inline int esz(int dtype) { return dtype == HIPT_F32 ? 4 : 2; }
inline hipStream_t S(void* s) { return (hipStream_t)s; }

struct Carver {
    char* base;
    size_t cap, used = 0;
    Carver(void* b, size_t c) : base((char*)b), cap(c) {}
    void* take(size_t n) {
        void* p = base + used;
        used += al256(n);
        return p;
    }
    bool ok() const { return used <= cap && (((uintptr_t)base & 255) == 0 || used == 0); }
};

// ---- optional per-kernel timing (bench.py's roofline leg): HIP events around every launch -----
enum { PC_EMBED, PC_LN, PC_QKV, PC_ATTN, PC_PROJ, PC_FC1, PC_FC2, PC_MLP, PC_ABMIL, PC_COMBINE, PC_OTHER, PC_VIT4K, PC_LASTCLS, PC_QKVATT, PC_CLSROWS, PC_N };
const char* const kProfNames[PC_N] = {"embed_gemm", "layernorm", "qkv_gemm", "attention", "proj_gemm",
                                      "fc1_gemm",   "fc2_gemm",  "mlp_fused",   "abmil_fused", "abmil_combine", "other",
                                      "vit4k_blocks", "last_block_cls", "qkv_attention_fused", "qkv_cls_rows"};
constexpr int kProfMax = 8192;
struct Prof {
    bool on = false, created = false;
    hipEvent_t ev[kProfMax][2];
    int cat[kProfMax];
    int n = 0;
} g_prof;

inline void prof_begin(int cat, hipStream_t st) {
    if (g_prof.on && g_prof.n < kProfMax) {
        g_prof.cat[g_prof.n] = cat;
        (void)hipEventRecord(g_prof.ev[g_prof.n][0], st);
    }
}
inline void prof_end(hipStream_t st) {
    if (g_prof.on && g_prof.n < kProfMax) {
        (void)hipEventRecord(g_prof.ev[g_prof.n][1], st);
        ++g_prof.n;
    }
}
#define PROF(cat, expr)        \
    do {                       \
        prof_begin(cat, st);   \
        rc = (expr);           \
        prof_end(st);          \
        if (rc) return rc;     \
    } while (0)

int check_vit(const hipt_vit_weights* w) {
    HIPT_CHECK_ARG(w != nullptr && w->blocks != nullptr, "vit: null weights");
    HIPT_CHECK_ARG(w->dtype == HIPT_F32 || w->dtype == HIPT_BF16, "vit: bad dtype %d", w->dtype);
    HIPT_CHECK_ARG(w->dim > 0 && w->dim % 64 == 0, "vit: dim=%d must be a multiple of 64", w->dim);
    HIPT_CHECK_ARG(w->heads > 0 && w->dim % w->heads == 0, "vit: dim %d not divisible by heads %d", w->dim, w->heads);
    const int dh = w->dim / w->heads;
    HIPT_CHECK_ARG(dh == 32 || dh == 64, "vit: head dim %d not in {32,64}", dh);
    HIPT_CHECK_ARG(w->hidden % 64 == 0, "vit: hidden=%d must be a multiple of 64", w->hidden);
    HIPT_CHECK_ARG(w->ln_eps > 0.f && w->ln_eps < 1.f, "vit: ln_eps=%g looks uninitialised", (double)w->ln_eps);
    if (w->ntok < 1 || w->ntok > 288) {
        hipt_set_error("vit: ntok=%d outside the on-chip attention envelope [1, 288]", w->ntok);
        return HIPT_E_UNSUPPORTED;
    }
    return HIPT_OK;
}

// Attention.scale (vision_transformer.py:112): qk_scale when the module was built with one, else head_dim ** -0.5
inline float attn_scale(const hipt_vit_weights* w) { return w->attn_scale > 0.f ? w->attn_scale : 1.0f / sqrtf((float)(w->dim / w->heads)); }

struct BlockScratch {
    void *xn, *qkv, *att, *hid;
};

size_t block_scratch_bytes(const hipt_vit_weights* w, int nseq) {
    const size_t rows = (size_t)nseq * w->ntok, e = esz(w->dtype);
    return al256(rows * w->dim * e) * 2 + al256(rows * 3 * w->dim * e) + al256(rows * w->hidden * e);
}

BlockScratch carve_blocks(Carver& c, const hipt_vit_weights* w, int nseq) {
    const size_t rows = (size_t)nseq * w->ntok, e = esz(w->dtype);
    BlockScratch s;
    s.xn = c.take(rows * w->dim * e);
    s.qkv = c.take(rows * 3 * w->dim * e);
    s.att = c.take(rows * w->dim * e);
    s.hid = c.take(rows * w->hidden * e);
    return s;
}

int linear(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* resid, void* out,
           int64_t ldc, int M, int N, int K, int dtype, int flags, hipStream_t st, int rpt = 0, const float* ln_w = nullptr,
           const float* ln_b = nullptr, float ln_eps = 0.f, int a_row_step = 0, int small_any = 0) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A;
    p.lda = lda;
    p.W = W;
    p.ldw = ldw;
    p.M = M;
    p.N = N;
    p.K = K;
    p.bias = bias;
    p.resid = resid;
    p.out = out;
    p.ldc = ldc;
    p.rpt = rpt;
    p.ln_w = ln_w;  // (set: A is the fp32 residual rows and the GEMM normalises them itself -- small calls, gemm.hip)
    p.ln_b = ln_b;
    p.ln_eps = ln_eps;
    p.a_row_step = a_row_step;  // (> 0: A is a bf16 activation image and row r is its row r * a_row_step)
    p.small_any = small_any;
    return hipt_gemm_launch(p, dtype, ALOAD_PLAIN, flags, st);
}

// Linears over ONE row per sequence (the [CLS] rows: M = nseq).  Up to 1 088 rows the small-M GEMM (gemm.hip: a wave per 16 x 32 output
// tile, reading the rows out of the activation image itself when a_row_step > 0), above that a gather launch (image rows only) + the
// tiled GEMM -- and BOTH walk k in the same ascending 32-element steps (GemmParams::asc), so that a sequence's bits do not depend
// on how many sequences share the call (feature_store.extract_slide gathers loader batches on that promise).
// a_row_step > 0: A is a bf16 activation image, GEMM row r its row r * a_row_step; `gather` = [M, K] scratch for the gathered rows.
int rows_linear(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out, int64_t ldc, int M, int N, int K, int dtype,
                hipStream_t st, int a_row_step = 0, void* gather = nullptr) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.W = W; p.ldw = ldw; p.M = M; p.N = N; p.K = K; p.bias = bias; p.out = out; p.ldc = ldc;
    p.asc = 1;
    if (a_row_step > 0) {
        if (!hipt_generic_only() && hipt_gemm_arows_supported(M, K, dtype, ALOAD_PLAIN, 0)) {
            p.a_row_step = a_row_step;
        } else {
            int rc = hipt_gather_cls_bf16_launch(A, gather, M, a_row_step, K, st, 1);
            if (rc) return rc;
            p.A = gather;
        }
    }
    return hipt_gemm_launch(p, dtype, ALOAD_PLAIN, 0, st);
}

// emit_last: the MLP of block b1-1 also writes LayerNorm-1 of block b1 on its output rows (bf16, s.att), for a caller
// that runs block b1 itself (the [CLS]-pruned last block); *have_xn tells it whether that happened
// img_ok / x_img_out: the caller owns x and accepts it back as an fp32 activation image (kernels.h): blocks after the
// first then exchange y1 / xn / x as images (coalesced row phases); *x_img_out tells whether x came back as one
// at most four 272-row sequences: what gemm.hip's small-M kernel takes (HIPT_GENERIC keeps its meaning: the generic kernels either way)
static bool small_call(const hipt_vit_weights* w, int nseq) { return (int64_t)nseq * w->ntok <= 1088; }

// do blocks [b0, b1) of a call of nseq sequences run LayerNorm-chained on the streaming kernels / exchange activation images?
static bool blocks_chain(const hipt_vit_weights* w, int nseq, int b0, int b1) {
    const int D = w->dim, dt = w->dtype;
    bool chain = hipt_seqgemm_supported(dt, D) && hipt_mlp_supported(dt, D, w->hidden) && !small_call(w, nseq) && !hipt_generic_only() &&
                 hipt_mlp16_supported(dt, D, w->hidden) && hipt_seqgemm_pipe_supported(dt, D, 3 * D, false, 0);
    for (int i = b0; i < b1 && chain; ++i) chain = w->blocks[i].qkv_pk && w->blocks[i].proj_pk && w->blocks[i].mlp_pk && (w->blocks[i].mlp_pk_fmt == 2 || w->blocks[i].mlp_pk_fmt == 3);
    return chain;
}
static bool blocks_images(const hipt_vit_weights* w, int nseq, int b0, int b1) {
    return blocks_chain(w, nseq, b0, b1) && ((int64_t)nseq * w->ntok) % 16 == 0 && !hipt_env_on("HIPT_NO_IMG") &&
           hipt_attention64_supported(w->dtype, w->dim / w->heads, w->ntok, false);
}

// xn_ready (round 5): the caller's embedding already left x as the fp32 activation image and LayerNorm-1 of block b0 as the bf16 image s.att
// (embed32.hip, LNOUT): block b0 then runs like every later block.  Only with img_ok and blocks_images(..) true (checked).
int run_blocks(const hipt_vit_weights* w, float* x, int nseq, int b0, int b1, float* probs, const BlockScratch& s,
               hipStream_t st, bool emit_last = false, bool* have_xn_out = nullptr, bool img_ok = false, bool* x_img_out = nullptr, bool xn_ready = false,
               bool force_small = false) {
    const int D = w->dim, M = nseq * w->ntok, dt = w->dtype, dh = D / w->heads;
    const float scale = attn_scale(w);
    int rc;
    // A call of a few hundred rows (ONE 256 x 256 patch: 257; the second-level ViT of one region: 257) is latency, not throughput: the
    // A-stationary kernels would put it on two 192-row tiles, the fused MLP on seventeen 16-row tiles that each stream the whole weight
    // image (55 us a block).  Such calls take the per-operator path below, whose Linears run on the small-M GEMM (gemm.hip: a wave per
    // 16 x 32 output tile, 51-204 workgroups): seven launches of a few microseconds per block.
    // force_small (hipt_vit4k_forward): the small-call kernels for ANY number of rows -- one launch per operator for all the regions of a call
    const bool small = small_call(w, nseq) || force_small;
    const int sa = force_small ? 1 : 0;
    const bool seq = hipt_seqgemm_supported(dt, D) && hipt_mlp_supported(dt, D, w->hidden) && !small;
    // timing categories: the kernels of the small second-level ViT (D = 192, a few hundred rows) are booked together,
    // so that the per-kernel categories hold only the ViT-256 launches the roofline is computed on
    const bool big = D >= 384;
    const int cQKV = big ? PC_QKV : PC_VIT4K, cATTN = big ? PC_ATTN : PC_VIT4K, cPROJ = big ? PC_PROJ : PC_VIT4K, cMLP = big ? PC_MLP : PC_VIT4K;
    // pipelined path: the MLP of block i applies LayerNorm-1 of block i+1 to the rows it finishes and leaves them in
    // s.att as bf16 operands; block i+1's QKV GEMM then skips the fp32 row load + LayerNorm
    // (only the streaming kernels have that epilogue / prologue: every block of the range needs its packed weight images)
    const bool chain = !force_small && blocks_chain(w, nseq, b0, b1);
    // activation images: chained streaming blocks, whole 16-row fragments, no probability output
    const bool img = !force_small && img_ok && probs == nullptr && blocks_images(w, nseq, b0, b1);
    HIPT_CHECK_ARG(!xn_ready || (img && b0 < b1), "run_blocks: image input without the image path");
    bool have_xn = xn_ready;
    // with them, q | k | v leave the QKV GEMM head-major (the attention kernel's K / V staging reads consecutive bytes)
    const bool hm = img && (int64_t)M * 3 * D * 2 < ((int64_t)1 << 32) - 65536;
    bool x_img = xn_ready;
    // the tile queues of the streaming kernels (three ints in the unused hidden slot) reset themselves at the end of a launch:
    // zeroed once here instead of before each of the ~44 launches (5 us each on the stream: 3 % of a one-region forward)
    const bool qz = chain && hipMemsetAsync(s.hid, 0, 48 * sizeof(int), st) == hipSuccess;
    for (int i = b0; i < b1; ++i) {
        const hipt_block_weights& b = w->blocks[i];
        const bool last_probs = probs != nullptr && i == b1 - 1;
        if (seq) {
            // bf16, D in {192,384}: A-stationary kernels.  QKV with LayerNorm-1 fused into the activation
            // load; proj leaves the attention-branch output y1 in bf16 (s.xn); the fused MLP kernel folds
            // y1 in, does LN2 + fc1 + GELU + fc2 with the hidden tensor on chip and updates x in place.
            SeqGemmParams q;
            memset(&q, 0, sizeof(q));
            q.M = M; q.K = D; q.ln_eps = w->ln_eps;
            q.A = x; q.lda = D; q.ln_w = b.ln1_w; q.ln_b = b.ln1_b; q.W = b.qkv_w; q.wpk = b.qkv_pk; q.N = 3 * D; q.bias = b.qkv_b;
            q.out = s.qkv; q.ldc = 3 * D;
            // (the hidden tensor is never materialised on this path: its slot holds the kernels' tile queues)
            q.counter = (int*)s.hid + 16;
            q.counter_zeroed = qz ? 1 : 0;
            q.out_ntok = w->ntok;
            // LayerNorm-chained block with activation images: the QKV projection runs inside the attention kernel (qkv_attention.hip),
            // q | k | v never reach HBM.  The [CLS] rows (257 = 8 x 32 + 1) get their q | k | v from a side GEMM over nseq rows.
            const bool fuse = have_xn && img && b.qkv_att_pk && hipt_qkv_attn_supported(dt, D, w->heads, w->ntok) && !last_probs &&
                              !hipt_env_on("HIPT_NO_FUSED_ATTN");
            // image format 3 (round 5): the output projection runs at the head of the fused MLP's tiles (mlp16.hip, FOLD) -- no proj launch, no y1
            const bool fold = chain && b.mlp_pk_fmt == 3 && !hipt_env_on("HIPT_NO_PROJ_FOLD") && !last_probs;
            // Where the attention output goes.  Fused kernel: the (unused) qkv slot.  Two kernels: s.att -- except under the fold, where the
            // MLP kernel reads the attention rows as y1 AND writes the next block's LayerNorm-1 rows to s.att in the same launch: the y1 slot
            // s.xn is free then (no proj launch writes it), so the two never share a buffer.  (The kernel itself would tolerate the alias -- a
            // workgroup loads all attention rows of its tile before it stores any, tiles own disjoint rows: mlp16.hip, "in place" -- but
            // nothing in a launch sequence should rest on that.)
            void* att_two = fold ? s.xn : s.att;
            const void* att_out = att_two;
            if (fuse) {
                char* qa = (char*)s.hid + 4096 + al256((size_t)nseq * D * 4);   // (the hidden slot is free on this path; its head holds tile queues)
                char* qcls = qa + al256((size_t)nseq * D * 2);                   // [nseq, 3 D] bf16 + 1 KiB the kernel's row DMA may read past the end
                // (nseq rows are a handful of the streaming kernel's 192-row tiles -- 11 CUs for 2 048 patches; the generic GEMM tiles N as well.
                //  Up to 1 088 sequences the small-M GEMM reads the [CLS] rows out of the image itself: one launch, not two)
                PROF(PC_CLSROWS, rows_linear(s.att, D, b.qkv_w, D, b.qkv_b, qcls, 3 * D, nseq, 3 * D, D, dt, st, w->ntok, qa));
                PROF(PC_QKVATT, hipt_qkv_attn_launch(s.att, b.qkv_att_pk, b.qkv_b, qcls, s.qkv, nseq, scale, st));
                att_out = s.qkv;
            } else if (have_xn) {  // LayerNorm-1 already applied by the previous block's MLP epilogue
                q.A = s.att; q.ln_w = q.ln_b = nullptr;
                q.img = (img ? 1 : 0) | (hm ? 4 : 0);
                PROF(cQKV, hipt_seqgemm_launch(q, false, 0, st));
            } else {
                q.img = hm ? 4 : 0;
                PROF(cQKV, hipt_seqgemm_launch(q, true, 0, st));
            }
            q.img = 0;
            // (with activation images the attention output is one too: proj then reads its operands 1 KiB at a time)
            if (!fuse) PROF(cATTN, hipt_attention_launch(s.qkv, att_two, last_probs ? probs : nullptr, nseq, w->ntok, w->heads, dh, scale, dt, st, img ? 1 : 0, hm ? 1 : 0));
            if (last_probs) break;
            if (!fold) {
                q.A = att_out; q.ln_w = q.ln_b = nullptr; q.W = b.proj_w; q.wpk = b.proj_pk; q.N = D; q.bias = b.proj_b; q.out = s.xn; q.ldc = D;
                q.counter = (int*)s.hid + 32;
                q.img = img ? 3 : 0;  // A = the attention output image, out = y1 image
                PROF(cPROJ, hipt_seqgemm_launch(q, false, 0, st));
            }
            MlpParams m;
            memset(&m, 0, sizeof(m));
            m.x = x; m.y1 = fold ? att_out : s.xn; m.ln_w = b.ln2_w; m.ln_b = b.ln2_b; m.ln_eps = w->ln_eps;
            m.fold = fold ? 1 : 0;
            m.bproj = b.proj_b;
            m.w1 = b.fc1_w; m.b1 = b.fc1_b; m.w2 = b.fc2_w; m.b2 = b.fc2_b; m.wpk = b.mlp_pk; m.wpk_fmt = b.mlp_pk_fmt; m.M = M; m.D = D; m.hidden = w->hidden;
            m.counter = (int*)s.hid;
            m.counter_zeroed = qz ? 1 : 0;
            have_xn = chain && (i + 1 < b1 || emit_last) && i + 1 < w->depth;
            if (have_xn) {
                m.ln_next_w = w->blocks[i + 1].ln1_w;
                m.ln_next_b = w->blocks[i + 1].ln1_b;
                m.xn_out = s.att;
            }
            if (img) {
                m.img = x_img ? 3 : 1;
                x_img = true;
            }
            PROF(cMLP, hipt_mlp_launch(m, st));
            continue;
        } else {
            // (small calls: both LayerNorms run in the prologue of the GEMM that consumes them -- five launches a block instead of seven)
            // (the kernel loads gamma / beta 16 bytes at a time: parameters that are views into a flat buffer off that grid keep the
            //  separate LayerNorm launch, which has no alignment requirement)
            const bool ln_al = (((uintptr_t)b.ln1_w | (uintptr_t)b.ln1_b | (uintptr_t)b.ln2_w | (uintptr_t)b.ln2_b) & 15) == 0;
            const bool ln_in_gemm = small && ln_al && !hipt_generic_only() && hipt_gemm_ln_supported(M, D, ALOAD_PLAIN, 0, force_small);
            if (ln_in_gemm) {
                PROF(PC_QKV, linear(x, D, b.qkv_w, D, b.qkv_b, nullptr, s.qkv, 3 * D, M, 3 * D, D, dt, 0, st, w->ntok, b.ln1_w, b.ln1_b, w->ln_eps, 0, sa));
            } else {
                PROF(PC_LN, hipt_layernorm_launch(x, D, b.ln1_w, b.ln1_b, s.xn, dt, D, M, D, w->ln_eps, st));
                PROF(PC_QKV, linear(s.xn, D, b.qkv_w, D, b.qkv_b, nullptr, s.qkv, 3 * D, M, 3 * D, D, dt, 0, st, w->ntok, nullptr, nullptr, 0.f, 0, sa));
            }
            PROF(PC_ATTN, hipt_attention_launch(s.qkv, s.att, last_probs ? probs : nullptr, nseq, w->ntok, w->heads, dh, scale, dt, st));
            if (last_probs) break;  // Block.forward(return_attention=True) returns before the residual (:148-149)
            PROF(PC_PROJ, linear(s.att, D, b.proj_w, D, b.proj_b, x, x, D, M, D, D, dt, HIPT_EPI_RESID | HIPT_EPI_OUT_F32, st, w->ntok, nullptr, nullptr, 0.f, 0, sa));
            if (ln_in_gemm) {
                PROF(PC_FC1, linear(x, D, b.fc1_w, D, b.fc1_b, nullptr, s.hid, w->hidden, M, w->hidden, D, dt, HIPT_EPI_GELU, st, w->ntok, b.ln2_w, b.ln2_b,
                                    w->ln_eps, 0, sa));
            } else {
                PROF(PC_LN, hipt_layernorm_launch(x, D, b.ln2_w, b.ln2_b, s.xn, dt, D, M, D, w->ln_eps, st));
                PROF(PC_FC1, linear(s.xn, D, b.fc1_w, D, b.fc1_b, nullptr, s.hid, w->hidden, M, w->hidden, D, dt, HIPT_EPI_GELU, st, w->ntok, nullptr, nullptr, 0.f, 0, sa));
            }
        }
        PROF(PC_FC2, linear(s.hid, w->hidden, b.fc2_w, w->hidden, b.fc2_b, x, x, D, M, D, w->hidden, dt,
                            HIPT_EPI_RESID | HIPT_EPI_OUT_F32, st, w->ntok, nullptr, nullptr, 0.f, 0, sa));
    }
    if (have_xn_out) *have_xn_out = have_xn;
    if (x_img_out) *x_img_out = x_img;
    return HIPT_OK;
}

// Last block when only the [CLS] row is consumed afterwards (ViT.forward returns norm(x)[:, 0], vision_transformer.py:248-253):
// K and V are needed for every token, everything after that only for token 0 of each sequence -- the attention of one
// query per (sequence, head), then proj / residual / MLP on nseq rows instead of nseq * ntok (allowed, and
// the pruned FLOP figure is the one the roofline uses).  Leaves the final residual rows compact in xc [nseq, D].
static bool can_prune_last(const hipt_vit_weights* w) {
    return w->dtype == HIPT_BF16 && w->dim == 384 && w->dim / w->heads == 64 && w->ntok <= 320 && hipt_seqgemm_supported(w->dtype, w->dim) &&
           hipt_mlp_supported(w->dtype, w->dim, w->hidden) && !hipt_env_on("HIPT_NO_PRUNE");
}

static int run_last_block_cls(const hipt_vit_weights* w, float* x, int nseq, const BlockScratch& s, float* xc, bool have_xn, bool x_img, hipStream_t st) {
    const int D = w->dim, M = nseq * w->ntok;
    const hipt_block_weights& b = w->blocks[w->depth - 1];
    int rc;
    SeqGemmParams q;
    memset(&q, 0, sizeof(q));
    q.M = M; q.K = D; q.ln_eps = w->ln_eps;
    q.A = x; q.lda = D; q.ln_w = b.ln1_w; q.ln_b = b.ln1_b; q.W = b.qkv_w; q.wpk = b.qkv_pk; q.N = 3 * D; q.bias = b.qkv_b;
    q.out = s.qkv; q.ldc = 3 * D;
    q.counter = (int*)s.hid + 16;
    const void* att_rows = nullptr;  // [nseq, D] bf16: the attention output of the [CLS] tokens
    if (have_xn) {  // LayerNorm-1 already applied by the previous block's MLP epilogue (bf16 operands in s.att)
        q.A = s.att; q.ln_w = q.ln_b = nullptr;
        q.img = x_img ? 1 : 0;  // (x and the operands s.att change layout together)
        const bool fuse = x_img && b.qkv_att_pk && hipt_qkv_attn_supported(w->dtype, D, w->heads, w->ntok) && !hipt_env_on("HIPT_NO_FUSED_ATTN");
        if (fuse) {
            // The fused kernel's [CLS]-only form: K and V of every token are computed per (patch, head) and consumed in place by the one
            // query of the patch; q | k | v of the [CLS] rows themselves from the side GEMM, as in the other blocks.  Output: the
            // attention rows of the [CLS] tokens, compact, in the (unused) qkv slot.
            char* qa = (char*)s.hid + 4096 + al256((size_t)nseq * D * 4);
            char* qcls = qa + al256((size_t)nseq * D * 2);
            PROF(PC_LASTCLS, hipt_gather_cls_bf16_launch(s.att, qa, nseq, w->ntok, D, st, 1));
            PROF(PC_LASTCLS, rows_linear(qa, D, b.qkv_w, D, b.qkv_b, qcls, 3 * D, nseq, 3 * D, D, w->dtype, st));
            q.img = 0;
            PROF(PC_LASTCLS, hipt_qkv_attn_cls_launch(s.att, b.qkv_att_pk, b.qkv_b, qcls, s.qkv, nseq, attn_scale(w), st));
            att_rows = s.qkv;
        } else {
            // Only token 0 of a sequence asks a question in this block: K and V for every row (columns 384.. of the QKV Linear: the
            // weight image of an N tile is the 98 304 bytes of its rows, so the tail of the image IS the [K; V] matrix), Q for the
            // [CLS] rows alone -- their operands gathered into the free hidden slot, a [nseq, 384] GEMM scattered to rows s * ntok
            bf16_t* qa = (bf16_t*)((char*)s.hid + 4096 + al256((size_t)nseq * D * 4));
            PROF(PC_LASTCLS, hipt_gather_cls_bf16_launch(s.att, qa, nseq, w->ntok, D, st, x_img ? 1 : 0));
            const size_t wq = (size_t)D * D * 2;
            q.W = (const char*)b.qkv_w + wq; q.wpk = b.qkv_pk ? (const char*)b.qkv_pk + wq : nullptr; q.N = 2 * D; q.bias = b.qkv_b + D;
            q.out = (bf16_t*)s.qkv + D;
            PROF(PC_LASTCLS, hipt_seqgemm_launch(q, false, 0, st));  // (booked apart: the QKV category holds full-size launches only)
            q.img = 0;
            q.M = nseq; q.A = qa; q.W = b.qkv_w; q.wpk = b.qkv_pk; q.N = D; q.bias = b.qkv_b; q.out = s.qkv; q.ldc = w->ntok * 3 * D;
            q.counter = (int*)s.hid + 32;
            PROF(PC_LASTCLS, hipt_seqgemm_launch(q, false, 0, st));
            q.M = M; q.ldc = 3 * D;
        }
        q.img = 0;
    } else {
        PROF(PC_QKV, hipt_seqgemm_launch(q, true, 0, st));
    }
    // (the [CLS]-row launches of the pruned block are booked apart: the per-kernel categories then hold full-size launches only)
    if (!att_rows) {
        PROF(PC_LASTCLS, hipt_attn_cls_launch(s.qkv, s.att, nullptr, nseq, w->ntok, w->heads, D / w->heads, attn_scale(w), st));
        att_rows = s.att;
    }
    PROF(PC_OTHER, hipt_gather_cls_launch(x, xc, nseq, (int64_t)w->ntok * D, D, st, x_img ? 1 : 0));
    // (nseq rows: the generic GEMM tiles N as well -- see the side GEMM of the fused blocks)
    PROF(PC_LASTCLS, rows_linear(att_rows, D, b.proj_w, D, b.proj_b, s.xn, D, nseq, D, D, w->dtype, st));
    MlpParams m;
    memset(&m, 0, sizeof(m));
    m.x = xc; m.y1 = s.xn; m.ln_w = b.ln2_w; m.ln_b = b.ln2_b; m.ln_eps = w->ln_eps;
    m.w1 = b.fc1_w; m.b1 = b.fc1_b; m.w2 = b.fc2_w; m.b2 = b.fc2_b; m.wpk = b.mlp_pk; m.wpk_fmt = b.mlp_pk_fmt; m.M = nseq; m.D = D; m.hidden = w->hidden;
    m.counter = (int*)s.hid;
    PROF(PC_LASTCLS, hipt_mlp_launch(m, st));
    return HIPT_OK;
}

int64_t image_elems(const hipt_image_layout* lay, int nseq_total) {
    const int per = lay->grid_w * lay->grid_h;
    return (int64_t)((nseq_total + per - 1) / per) * lay->batch_stride;
}

// tokens of sequences [seq0, seq0+nseq) from an image tensor already in the compute dtype
int embed256(const hipt_vit_weights* w, const void* img, const hipt_image_layout* lay, int seq0, int nseq, float* x,
             hipStream_t st) {
    HIPT_CHECK_ARG(lay->patch_h % 16 == 0 && lay->patch_w % 16 == 0, "vit256: patch %dx%d not a multiple of 16", lay->patch_h,
                   lay->patch_w);
    const int nty = lay->patch_h / 16, ntx = lay->patch_w / 16;
    HIPT_CHECK_ARG(nty * ntx + 1 == w->ntok, "vit256: image gives %d tokens, weights expect %d", nty * ntx + 1, w->ntok);
    HIPT_CHECK_ARG(w->embed_k == 768, "vit256: embed_k must be 768 (3x16x16)");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = img;
    p.W = w->embed_w;
    p.ldw = w->embed_k;
    p.M = nseq * nty * ntx;
    p.N = w->dim;
    p.K = w->embed_k;
    p.bias = w->embed_b;
    p.out = x;
    p.ldc = w->dim;
    p.pos = w->pos;
    p.rows_per_seq = nty * ntx;
    p.rpt = nty * ntx;
    p.im = *lay;
    p.im_nty = nty;
    p.im_ntx = ntx;
    p.im_seq0 = seq0;
    int rc;
    PROF(PC_EMBED, hipt_gemm_launch(p, w->dtype, ALOAD_IM2COL, HIPT_EPI_OUT_F32 | EPI_ROWMAP, st));
    PROF(PC_OTHER, hipt_cls_init_launch(x, w->cls, w->pos, nseq, w->ntok, w->dim, st));
    return HIPT_OK;
}

// may the embedding read fp32 pixels itself (embed32.hip)?  `slot` bytes are available for its packed weight + tile queue
static bool embed_fused_ok(const hipt_vit_weights* w, const void* images, const hipt_image_layout* lay, size_t slot, int kind = 0) {
    // (uint8: 8-byte pixel runs; interleaved tensors are whole [n, W, H, 3] images: batch_stride = 3 * chan_stride)
    if (kind != 0 && (lay->row_stride % 8 != 0 || lay->chan_stride % 8 != 0 || lay->batch_stride != 3 * lay->chan_stride))
        return false;
    return lay->patch_w % 16 == 0 && lay->patch_h % 16 == 0 &&
           hipt_embed32_supported(w->dtype, w->dim, w->embed_k, lay->patch_h / 16, lay->patch_w / 16) &&
           w->ntok == (lay->patch_h / 16) * (lay->patch_w / 16) + 1 && slot >= hipt_embed32_packed_bytes() + 256 && ((uintptr_t)images % 16) == 0 &&
           lay->row_stride % 4 == 0 && lay->chan_stride % 4 == 0 && lay->batch_stride % 4 == 0;
}

// the same from the fp32 image itself (embed32.hip): `wpk` = the packed Conv2d weight, `counter` = the kernel's tile queue
// xn_img != null: x leaves as the fp32 activation image and LayerNorm-1 of the first block as the bf16 image xn_img (run_blocks: xn_ready)
int embed256_f32(const hipt_vit_weights* w, const void* img, const hipt_image_layout* lay, int seq0, int nseq, float* x, const void* wpk,
                 int* counter, hipStream_t st, int kind = 0, void* xn_img = nullptr) {
    EmbedParams p;
    memset(&p, 0, sizeof(p));
    p.img = img;
    p.kind = kind;
    p.im = *lay;
    p.nty = lay->patch_h / 16;
    p.ntx = lay->patch_w / 16;
    p.seq0 = seq0;
    p.nseq = nseq;
    p.wpk = wpk;
    p.bias = w->embed_b;
    p.pos = w->pos;
    p.x = x;
    p.ntok = w->ntok;
    p.counter = counter;
    int rc;
    if (xn_img) {
        p.xn_out = xn_img;
        p.ln_w = w->blocks[0].ln1_w;
        p.ln_b = w->blocks[0].ln1_b;
        p.ln_eps = w->ln_eps;
        PROF(PC_EMBED, hipt_embed32_launch(p, st));
        PROF(PC_OTHER, hipt_cls_init_img_launch(x, xn_img, w->cls, w->pos, p.ln_w, p.ln_b, p.ln_eps, nseq, w->ntok, w->dim, st));
        return HIPT_OK;
    }
    PROF(PC_EMBED, hipt_embed32_launch(p, st));
    PROF(PC_OTHER, hipt_cls_init_launch(x, w->cls, w->pos, nseq, w->ntok, w->dim, st));
    return HIPT_OK;
}

int embed4k(const hipt_vit_weights* w, const void* tokens, int nseq, float* x, hipStream_t st, int small_any = 0) {
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.small_any = small_any;
    p.A = tokens;
    p.lda = w->embed_k;
    p.W = w->embed_w;
    p.ldw = w->embed_k;
    p.M = nseq * (w->ntok - 1);
    p.N = w->dim;
    p.K = w->embed_k;
    p.bias = w->embed_b;
    p.out = x;
    p.ldc = w->dim;
    p.pos = w->pos;
    p.rows_per_seq = w->ntok - 1;
    p.rpt = w->ntok - 1;
    int rc = HIPT_OK;
    if (p.M > 0) PROF(PC_EMBED, hipt_gemm_launch(p, w->dtype, ALOAD_PLAIN, HIPT_EPI_GELU | HIPT_EPI_OUT_F32 | EPI_ROWMAP, st));
    PROF(PC_OTHER, hipt_cls_init_launch(x, w->cls, w->pos, nseq, w->ntok, w->dim, st));
    return HIPT_OK;
}

int default_chunk(int nseq) { return nseq < 2048 ? nseq : 2048; }

// input image kinds: fp32 [.., 3, W, H], or uint8 in the same layout / interleaved [.., W, H, 3] (normalised on device)
enum { IMG_F32 = 0, IMG_U8_CHW = 1, IMG_U8_HWC = 2 };

}  // namespace

extern "C" {

int hipt_abi_version(void) { return HIPT_ABI_VERSION; }

int hipt_profile_enable(int on) {
    if (on && !g_prof.created) {
        for (int i = 0; i < kProfMax; ++i)
            for (int j = 0; j < 2; ++j)
                if (hipEventCreate(&g_prof.ev[i][j]) != hipSuccess) {
                    hipt_set_error("profile: hipEventCreate failed");
                    return HIPT_E_LAUNCH;
                }
        g_prof.created = true;
    }
    g_prof.on = on != 0;
    g_prof.n = 0;
    return HIPT_OK;
}
int hipt_profile_categories(void) { return PC_N; }
const char* hipt_profile_category_name(int i) { return (i >= 0 && i < PC_N) ? kProfNames[i] : ""; }
int hipt_profile_read(float* ms, int* counts) {
    for (int i = 0; i < PC_N; ++i) {
        ms[i] = 0.f;
        counts[i] = 0;
    }
    for (int i = 0; i < g_prof.n; ++i) {
        if (hipEventSynchronize(g_prof.ev[i][1]) != hipSuccess) {
            hipt_set_error("profile: hipEventSynchronize failed");
            return HIPT_E_LAUNCH;
        }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, g_prof.ev[i][0], g_prof.ev[i][1]);
        ms[g_prof.cat[i]] += t;
        counts[g_prof.cat[i]] += 1;
    }
    const int dropped = g_prof.n >= kProfMax;
    g_prof.n = 0;
    return dropped ? HIPT_E_WORKSPACE : HIPT_OK;
}
const char* hipt_last_error(void) { return g_err; }

int hipt_layernorm(const float* x, int64_t x_stride, const float* w, const float* b, void* out, int out_dtype,
                   int64_t out_stride, int rows, int D, float eps, void* stream) {
    return hipt_layernorm_launch(x, x_stride, w, b, out, out_dtype, out_stride, rows, D, eps, S(stream));
}

int hipt_linear(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* resid, void* out,
                int64_t ldc, int M, int N, int K, int dtype, int flags, void* stream) {
    HIPT_CHECK_ARG((flags & ~(HIPT_EPI_GELU | HIPT_EPI_RESID | HIPT_EPI_OUT_F32 | HIPT_EPI_RELU)) == 0, "linear: bad flags %d",
                   flags);
    HIPT_CHECK_ARG(!(flags & HIPT_EPI_RESID) || resid != nullptr, "linear: RESID without a residual pointer");
    return linear(A, lda, W, ldw, bias, resid, out, ldc, M, N, K, dtype, flags, S(stream));
}

int hipt_attention(const void* qkv, void* out, float* probs, int B, int ntok, int heads, int dh, float scale, int dtype,
                   void* stream) {
    return hipt_attention_launch(qkv, out, probs, B, ntok, heads, dh, scale, dtype, S(stream));
}

size_t hipt_vit_workspace_bytes(const hipt_vit_weights* w, int nseq) {
    // block scratch; the 4K token conversion buffer is the only extra of the prepare_tokens calls
    return block_scratch_bytes(w, nseq) + al256((size_t)nseq * w->ntok * w->embed_k * 2);
}

size_t hipt_vit256_forward_workspace_bytes(const hipt_vit_weights* w, const hipt_image_layout* lay, int nseq, int chunk) {
    if (chunk <= 0) chunk = default_chunk(nseq);
    if (chunk > nseq) chunk = nseq;
    size_t n = al256((size_t)chunk * w->ntok * w->dim * 4) + block_scratch_bytes(w, chunk);
    if (w->dtype == HIPT_BF16) n += al256((size_t)image_elems(lay, nseq) * 2);
    return n;
}

size_t hipt_vit4k_forward_workspace_bytes(const hipt_vit_weights* w, int nseq) {
    return al256((size_t)nseq * w->ntok * w->dim * 4) + hipt_vit_workspace_bytes(w, nseq);
}

int hipt_vit256_prepare_tokens(const hipt_vit_weights* w, const float* images, const hipt_image_layout* lay, int seq0,
                               int nseq, float* x, void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(images && lay && x && nseq > 0 && seq0 >= 0, "vit256_prepare_tokens: null/empty argument");
    const void* img = images;
    if (w->dtype == HIPT_BF16) {
        const int64_t n = image_elems(lay, seq0 + nseq);
        if (ws_bytes < al256((size_t)n * 2) || ((uintptr_t)workspace & 255)) {
            hipt_set_error("vit256_prepare_tokens: workspace %zu B too small / unaligned (need %zu)", ws_bytes, al256((size_t)n * 2));
            return HIPT_E_WORKSPACE;
        }
        if (embed_fused_ok(w, images, lay, ws_bytes)) {
            if ((rc = hipt_embed32_pack_launch(w->embed_w, workspace, S(stream)))) return rc;
            return embed256_f32(w, images, lay, seq0, nseq, x, workspace, (int*)((char*)workspace + hipt_embed32_packed_bytes()), S(stream));
        }
        if ((rc = hipt_f32_to_bf16_launch(images, workspace, n, S(stream)))) return rc;
        img = workspace;
    }
    return embed256(w, img, lay, seq0, nseq, x, S(stream));
}

int hipt_vit4k_prepare_tokens(const hipt_vit_weights* w, const float* tokens_in, int nseq, float* x, void* workspace,
                              size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(x && nseq > 0 && (tokens_in || w->ntok == 1), "vit4k_prepare_tokens: null/empty argument");
    const void* tok = tokens_in;
    const int64_t n = (int64_t)nseq * (w->ntok - 1) * w->embed_k;
    if (w->dtype == HIPT_BF16 && n > 0) {
        if (ws_bytes < al256((size_t)n * 2) || ((uintptr_t)workspace & 255)) {
            hipt_set_error("vit4k_prepare_tokens: workspace %zu B too small / unaligned (need %zu)", ws_bytes, al256((size_t)n * 2));
            return HIPT_E_WORKSPACE;
        }
        if ((rc = hipt_f32_to_bf16_launch(tokens_in, workspace, n, S(stream)))) return rc;
        tok = workspace;
    }
    return embed4k(w, tok, nseq, x, S(stream));
}

int hipt_vit_blocks(const hipt_vit_weights* w, float* x, int nseq, int blk_begin, int blk_end, float* probs, void* workspace,
                    size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(x && nseq > 0 && blk_begin >= 0 && blk_end <= w->depth && blk_begin <= blk_end, "vit_blocks: bad range [%d,%d)",
                   blk_begin, blk_end);
    Carver c(workspace, ws_bytes);
    BlockScratch s = carve_blocks(c, w, nseq);
    if (!c.ok()) {
        hipt_set_error("vit_blocks: workspace %zu B too small / unaligned (need %zu)", ws_bytes, c.used);
        return HIPT_E_WORKSPACE;
    }
    return run_blocks(w, x, nseq, blk_begin, blk_end, probs, s, S(stream));
}

// the [CLS] row of the last block's attention map, probs_cls[nseq, heads, ntok], without the
// [nseq, heads, ntok, ntok] tensor (heat-maps read attention[:, :, 0, 1:], hipt_4k.py:143-158).  x = prepared tokens
// (modified: it ends as the input of the last block).  Every dtype / head dim check_vit admits.
int hipt_vit_cls_attention(const hipt_vit_weights* w, float* x, int nseq, float* probs_cls, void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(x && probs_cls && nseq > 0, "vit_cls_attention: null/empty argument");
    const int D = w->dim, dh = D / w->heads, M = nseq * w->ntok;
    Carver c(workspace, ws_bytes);
    BlockScratch s = carve_blocks(c, w, nseq);
    if (!c.ok()) {
        hipt_set_error("vit_cls_attention: workspace %zu B too small / unaligned (need %zu)", ws_bytes, c.used);
        return HIPT_E_WORKSPACE;
    }
    hipStream_t st = S(stream);
    const hipt_block_weights& b = w->blocks[w->depth - 1];
    if (w->dtype == HIPT_BF16 && dh == 64 && hipt_seqgemm_supported(w->dtype, D)) {  // ViT-256 hot case: chained kernels
        bool have_xn = false;
        if ((rc = run_blocks(w, x, nseq, 0, w->depth - 1, nullptr, s, st, true, &have_xn))) return rc;
        SeqGemmParams q;
        memset(&q, 0, sizeof(q));
        q.M = M; q.K = D; q.ln_eps = w->ln_eps;
        q.A = x; q.lda = D; q.ln_w = b.ln1_w; q.ln_b = b.ln1_b; q.W = b.qkv_w; q.wpk = b.qkv_pk; q.N = 3 * D; q.bias = b.qkv_b;
        q.out = s.qkv; q.ldc = 3 * D;
        q.counter = (int*)s.hid + 16;
        if (have_xn) {
            q.A = s.att; q.ln_w = q.ln_b = nullptr;
        }
        if ((rc = hipt_seqgemm_launch(q, !have_xn, 0, st))) return rc;
        return hipt_attn_cls_launch(s.qkv, nullptr, probs_cls, nseq, w->ntok, w->heads, dh, attn_scale(w), st);
    }
    // every other configuration (fp32; head dim 32 = ViT-4K): the blocks before the last, then LayerNorm-1 + the QKV
    // projection of the last one and the probabilities of the [CLS] query from the one-query kernel
    if ((rc = run_blocks(w, x, nseq, 0, w->depth - 1, nullptr, s, st))) return rc;
    if ((rc = hipt_layernorm_launch(x, D, b.ln1_w, b.ln1_b, s.xn, w->dtype, D, M, D, w->ln_eps, st))) return rc;
    if ((rc = linear(s.xn, D, b.qkv_w, D, b.qkv_b, nullptr, s.qkv, 3 * D, M, 3 * D, D, w->dtype, 0, st, w->ntok))) return rc;
    return hipt_attn_cls_probs_launch(s.qkv, probs_cls, nseq, w->ntok, w->heads, dh, attn_scale(w), w->dtype, st);
}

int hipt_vit_attention_unit(const hipt_vit_weights* w, int block, const void* xn_img, int nseq, void* out_img, int fused, void* workspace,
                            size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(xn_img && out_img && xn_img != out_img && nseq > 0 && block >= 0 && block < w->depth, "vit_attention_unit: bad argument");
    const int D = w->dim, M = nseq * w->ntok;
    if (!hipt_qkv_attn_supported(w->dtype, D, w->heads, w->ntok) || M % 16 != 0) {
        hipt_set_error("vit_attention_unit: bf16, D = 384, 6 heads, 257 tokens and nseq * 257 %% 16 == 0 only");
        return HIPT_E_UNSUPPORTED;
    }
    Carver c(workspace, ws_bytes);
    BlockScratch s = carve_blocks(c, w, nseq);
    if (!c.ok()) {
        hipt_set_error("vit_attention_unit: workspace %zu B too small / unaligned (need %zu)", ws_bytes, c.used);
        return HIPT_E_WORKSPACE;
    }
    hipStream_t st = S(stream);
    const hipt_block_weights& b = w->blocks[block];
    SeqGemmParams q;
    memset(&q, 0, sizeof(q));
    q.K = D; q.lda = D; q.W = b.qkv_w; q.wpk = b.qkv_pk; q.N = 3 * D; q.bias = b.qkv_b; q.ldc = 3 * D; q.out_ntok = w->ntok;
    q.counter = (int*)s.hid;
    HIPT_CHECK_ARG(b.qkv_pk != nullptr, "vit_attention_unit: blocks[%d].qkv_pk is NULL", block);
    if (fused) {
        HIPT_CHECK_ARG(b.qkv_att_pk != nullptr, "vit_attention_unit: blocks[%d].qkv_att_pk is NULL", block);
        char* qa = (char*)s.hid + 4096;
        char* qcls = qa + al256((size_t)nseq * D * 2);
        if ((rc = hipt_gather_cls_bf16_launch(xn_img, qa, nseq, w->ntok, D, st, 1))) return rc;
        q.M = nseq; q.A = qa; q.out = qcls;
        if ((rc = hipt_seqgemm_launch(q, false, 0, st))) return rc;
        return hipt_qkv_attn_launch(xn_img, b.qkv_att_pk, b.qkv_b, qcls, out_img, nseq, attn_scale(w), st);
    }
    const bool hm = (int64_t)M * 3 * D * 2 < ((int64_t)1 << 32) - 65536;
    q.M = M; q.A = xn_img; q.out = s.qkv; q.img = 1 | (hm ? 4 : 0);
    if ((rc = hipt_seqgemm_launch(q, false, 0, st))) return rc;
    return hipt_attention_launch(s.qkv, out_img, nullptr, nseq, w->ntok, w->heads, D / w->heads, attn_scale(w), w->dtype, st, 1, hm ? 1 : 0);
}

int hipt_vit_mlp_unit(const hipt_vit_weights* w, int block, float* x_img, const void* att_img, int nseq, void* xn_out_img, void* workspace,
                      size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(x_img && att_img && nseq > 0 && block >= 0 && block < w->depth, "vit_mlp_unit: bad argument");
    const int D = w->dim, M = nseq * w->ntok;
    const hipt_block_weights& b = w->blocks[block];
    if (!hipt_mlp16_supported(w->dtype, D, w->hidden) || M % 16 != 0 || !b.mlp_pk || b.mlp_pk_fmt != 3) {
        hipt_set_error("vit_mlp_unit: bf16, D = 384, hidden %% 128 == 0, nseq * ntok %% 16 == 0 and blocks[%d].mlp_pk in format 3 only", block);
        return HIPT_E_UNSUPPORTED;
    }
    if (ws_bytes < 256 || ((uintptr_t)workspace & 255) || !workspace) {
        hipt_set_error("vit_mlp_unit: workspace %zu B too small / unaligned (need 256)", ws_bytes);
        return HIPT_E_WORKSPACE;
    }
    MlpParams m;
    memset(&m, 0, sizeof(m));
    m.x = x_img; m.y1 = att_img; m.fold = 1; m.bproj = b.proj_b;
    m.ln_w = b.ln2_w; m.ln_b = b.ln2_b; m.ln_eps = w->ln_eps;
    m.w1 = b.fc1_w; m.b1 = b.fc1_b; m.w2 = b.fc2_w; m.b2 = b.fc2_b; m.wpk = b.mlp_pk; m.wpk_fmt = b.mlp_pk_fmt; m.M = M; m.D = D; m.hidden = w->hidden;
    m.counter = (int*)workspace;
    m.img = 3;
    if (xn_out_img) {
        const hipt_block_weights& nb = w->blocks[block + 1 < w->depth ? block + 1 : block];
        m.ln_next_w = nb.ln1_w; m.ln_next_b = nb.ln1_b; m.xn_out = xn_out_img;
    }
    hipStream_t st = S(stream);
    PROF(PC_MLP, hipt_mlp_launch(m, st));
    return HIPT_OK;
}

// Format of the fused MLP's weight image: 2 = csrc/mlp16.hip (16x16x32 MFMAs), 0 = this shape has no packed form.  (Format 1 was the
// 32x32x16 form of rounds 2-4, tools/experiments/mlp32_r4.hip: a tie on the MLP launches themselves, 3 % behind on the kernels that run
// between them -- DESIGN.md -- and retired in round 5; an image packed as format 1 is refused by the chain test in run_blocks and its
// model falls back to the generic kernels.)
// 3 (round 5, what this version packs) = format 2 behind six units of the proj matrix: the attention block's output projection then runs at the
// head of the fused MLP's tiles (mlp16.hip, FOLD) and the chained blocks have no proj launch.  HIPT_NO_PROJ_FOLD=1 at LAUNCH time runs proj as its
// own kernel again from the same image (the MLP's own units lie behind the proj units); an image packed as format 2 by an older binding still runs.
int hipt_vit_mlp_pack_format(const hipt_vit_weights* w) { return w && hipt_mlp16_supported(w->dtype, w->dim, w->hidden) ? 3 : 0; }

size_t hipt_vit_packed_bytes(const hipt_vit_weights* w, int what) {
    if (!w || w->dtype != HIPT_BF16) return 0;
    const int D = w->dim;
    switch (what) {
        case HIPT_PACK_QKV: return hipt_seqgemm_pipe_supported(w->dtype, D, 3 * D, false, 0) ? (size_t)3 * D * D * 2 : 0;
        case HIPT_PACK_PROJ: return hipt_seqgemm_pipe_supported(w->dtype, D, D, false, 0) ? (size_t)D * D * 2 : 0;
        case HIPT_PACK_MLP: return hipt_vit_mlp_pack_format(w) != 0 ? hipt_mlp16_packed_bytes(D, w->hidden, hipt_vit_mlp_pack_format(w) == 3) : 0;
        case HIPT_PACK_QKV_ATT: return hipt_qkv_attn_supported(w->dtype, D, w->heads, w->ntok) ? hipt_qkv_attn_packed_bytes() : 0;
        default: return 0;
    }
}

int hipt_vit_pack_weights(const hipt_vit_weights* w, int block, int what, void* out, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(block >= 0 && block < w->depth, "vit_pack_weights: block %d of %d", block, w->depth);
    HIPT_CHECK_ARG(out != nullptr, "vit_pack_weights: null output");
    if (hipt_vit_packed_bytes(w, what) == 0) {
        hipt_set_error("vit_pack_weights: matrix %d of this ViT (dtype %d, D=%d, hidden=%d) has no packed form", what, w->dtype, w->dim, w->hidden);
        return HIPT_E_UNSUPPORTED;
    }
    const hipt_block_weights& b = w->blocks[block];
    const int D = w->dim;
    hipStream_t st = S(stream);
    switch (what) {
        case HIPT_PACK_QKV: return hipt_seqgemm_pack_launch(b.qkv_w, 3 * D, D, out, st);
        case HIPT_PACK_PROJ: return hipt_seqgemm_pack_launch(b.proj_w, D, D, out, st);
        case HIPT_PACK_QKV_ATT: return hipt_qkv_attn_pack_launch(b.qkv_w, out, st);
        default:
            // the format the caller recorded beside the pointer (hipt_vit_mlp_pack_format): pack and launch read the same field
            if ((b.mlp_pk_fmt == 2 || b.mlp_pk_fmt == 3) && hipt_mlp16_supported(w->dtype, D, w->hidden))
                return hipt_mlp16_pack_launch(b.fc1_w, b.fc2_w, D, w->hidden, out, st, b.mlp_pk_fmt == 3 ? b.proj_w : nullptr);
            hipt_set_error("hipt_vit_pack_weights: blocks[%d].mlp_pk_fmt = %d is not a format this model has", block, b.mlp_pk_fmt);
            return HIPT_E_BADARG;
    }
}

int hipt_vit_head(const hipt_vit_weights* w, const float* x, int nseq, int cls_only, float* out, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    const int D = w->dim;
    if (cls_only)
        return hipt_layernorm_launch(x, (int64_t)w->ntok * D, w->norm_w, w->norm_b, out, HIPT_F32, D, nseq, D, w->ln_eps, S(stream));
    return hipt_layernorm_launch(x, D, w->norm_w, w->norm_b, out, HIPT_F32, D, nseq * w->ntok, D, w->ln_eps, S(stream));
}

// images: fp32 [.., 3, W, H] (kind 0), or uint8 in the same layout (kind 1) / interleaved [.., W, H, 3] (kind 2), which
// are normalised on device into the compute dtype

static size_t image_extra_bytes(const hipt_vit_weights* w, const hipt_image_layout* lay, int nseq, int kind) {
    // bf16 mode already holds a bf16 image in its workspace; fp32 mode needs an fp32 one for uint8 input
    return (kind != IMG_F32 && w->dtype == HIPT_F32) ? al256((size_t)image_elems(lay, nseq) * 4) : 0;
}

// ViT-256 over the sequences [seq0, seq0 + nseq) of an image tensor that is ALREADY in the compute dtype, chunk by chunk:
// out[i] = [CLS] feature of sequence seq0 + i.  Scratch: the residual stream of one chunk + its block scratch.
// (embed_pk != null: `img` is the fp32 image and the embedding reads it directly -- embed32.hip; the tile queue sits behind the image)
static int vit256_range_impl(const hipt_vit_weights* w, const void* img, const hipt_image_layout* lay, int seq0, int nseq, int chunk, float* out,
                             void* workspace, size_t ws_bytes, hipStream_t st, const void* embed_pk = nullptr, int embed_kind = 0) {
    int rc;
    if (chunk <= 0) chunk = default_chunk(nseq);
    if (chunk > nseq) chunk = nseq;
    Carver c(workspace, ws_bytes);
    float* x = (float*)c.take((size_t)chunk * w->ntok * w->dim * 4);
    BlockScratch s = carve_blocks(c, w, chunk);
    if (!c.ok()) {
        hipt_set_error("vit256_forward: workspace %zu B too small / unaligned (need %zu)", ws_bytes, c.used);
        return HIPT_E_WORKSPACE;
    }
    for (int s0 = 0; s0 < nseq; s0 += chunk) {
        const int n = nseq - s0 < chunk ? nseq - s0 : chunk;
        const bool prune = can_prune_last(w) && !small_call(w, n);
        // the embedding hands the first block its operands as activation images when the blocks exchange images anyway (HIPT_NO_EMBED_LN: off)
        const bool pre = embed_pk && prune && w->depth > 1 && blocks_images(w, n, 0, w->depth - 1) && !hipt_env_on("HIPT_NO_EMBED_LN");
        if (embed_pk) {
            if ((rc = embed256_f32(w, img, lay, seq0 + s0, n, x, embed_pk, (int*)((char*)embed_pk + hipt_embed32_packed_bytes()), st, embed_kind, pre ? s.att : nullptr)))
                return rc;
        } else if ((rc = embed256(w, img, lay, seq0 + s0, n, x, st))) {
            return rc;
        }
        if (prune) {
            float* xc = (float*)((char*)s.hid + 4096);  // (the hidden-tensor slot is free on this path; its head holds tile queues)
            bool have_xn = false, x_img = false;  // (x is this function's own buffer: it may come back as an activation image)
            if ((rc = run_blocks(w, x, n, 0, w->depth - 1, nullptr, s, st, true, &have_xn, true, &x_img, pre))) return rc;
            if ((rc = run_last_block_cls(w, x, n, s, xc, have_xn, x_img, st))) return rc;
            PROF(PC_LN, hipt_layernorm_launch(xc, w->dim, w->norm_w, w->norm_b, out + (size_t)s0 * w->dim, HIPT_F32, w->dim, n, w->dim,
                                              w->ln_eps, st));
        } else {
            if ((rc = run_blocks(w, x, n, 0, w->depth, nullptr, s, st))) return rc;
            PROF(PC_LN, hipt_layernorm_launch(x, (int64_t)w->ntok * w->dim, w->norm_w, w->norm_b, out + (size_t)s0 * w->dim, HIPT_F32,
                                              w->dim, n, w->dim, w->ln_eps, st));
        }
    }
    return HIPT_OK;
}

static size_t vit256_range_bytes(const hipt_vit_weights* w, int nseq, int chunk) {
    if (chunk <= 0) chunk = default_chunk(nseq);
    if (chunk > nseq) chunk = nseq;
    return al256((size_t)chunk * w->ntok * w->dim * 4) + block_scratch_bytes(w, chunk);
}

// the input image tensor in the compute dtype: fp32 input in fp32 mode is used where it lies (returns `images`), everything
// else is converted / normalised into `dst`
static int image_to_compute(const hipt_vit_weights* w, const void* images, int kind, const hipt_image_layout* lay, int nseq, void* dst,
                            const void** img_out, hipStream_t st) {
    int rc;
    const int64_t n_img = image_elems(lay, nseq);
    if (kind != IMG_F32) {
        const int per = lay->grid_w * lay->grid_h;
        HIPT_CHECK_ARG(nseq % per == 0, "vit256_forward: uint8 input must hold whole regions");
        const int64_t plane = lay->batch_stride / 3;
        PROF(PC_OTHER, hipt_u8_normalize_launch(images, kind == IMG_U8_HWC, nseq / per, plane, dst, w->dtype, st));
        *img_out = dst;
    } else if (w->dtype == HIPT_BF16) {
        PROF(PC_OTHER, hipt_f32_to_bf16_launch((const float*)images, dst, n_img, st));
        *img_out = dst;
    } else {
        *img_out = images;
    }
    return HIPT_OK;
}

static int vit256_forward_impl(const hipt_vit_weights* w, const void* images, int kind, const hipt_image_layout* lay, int nseq, int chunk,
                               float* out, void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(images && lay && out && nseq > 0, "vit256_forward: null/empty argument");
    hipStream_t st = S(stream);
    const size_t nrange = vit256_range_bytes(w, nseq, chunk);
    const int64_t n_img = image_elems(lay, nseq);
    const size_t nimg = w->dtype == HIPT_BF16 ? al256((size_t)n_img * 2) : kind != IMG_F32 ? al256((size_t)n_img * 4) : 0;
    if (ws_bytes < nrange + nimg || ((uintptr_t)workspace & 255)) {
        hipt_set_error("vit256_forward: workspace %zu B too small / unaligned (need %zu)", ws_bytes, nrange + nimg);
        return HIPT_E_WORKSPACE;
    }
    const void* img = images;
    // fp32 pixels, bf16 model, 256 x 256 patches: the embedding kernel reads the image itself; the slot of the bf16 copy holds its
    // packed weight (made here: 0.6 MB, a few microseconds) and its tile queue instead
    // (uint8 RGB, planar or interleaved: the same kernel normalises in registers -- no device copy of the image at all)
    if (embed_fused_ok(w, images, lay, nimg, kind)) {
        void* pk = (char*)workspace + nrange;
        if ((rc = hipt_embed32_pack_launch(w->embed_w, pk, st))) return rc;
        return vit256_range_impl(w, images, lay, 0, nseq, chunk, out, workspace, nrange, st, pk, kind);
    }
    if ((rc = image_to_compute(w, images, kind, lay, nseq, (char*)workspace + nrange, &img, st))) return rc;
    return vit256_range_impl(w, img, lay, 0, nseq, chunk, out, workspace, nrange, st);
}

int hipt_vit256_forward(const hipt_vit_weights* w, const float* images, const hipt_image_layout* lay, int nseq, int chunk,
                        float* out, void* workspace, size_t ws_bytes, void* stream) {
    return vit256_forward_impl(w, images, IMG_F32, lay, nseq, chunk, out, workspace, ws_bytes, stream);
}

size_t hipt_image_compute_bytes(const hipt_vit_weights* w, const hipt_image_layout* lay, int nseq, int input_kind) {
    if (!w || !lay || nseq <= 0) return 0;
    const int64_t n_img = image_elems(lay, nseq);
    return w->dtype == HIPT_BF16 ? al256((size_t)n_img * 2) : input_kind != IMG_F32 ? al256((size_t)n_img * 4) : 0;
}

int hipt_image_to_compute(const hipt_vit_weights* w, const void* images, int input_kind, const hipt_image_layout* lay, int nseq, void* dst,
                          void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(images && lay && nseq > 0 && input_kind >= IMG_F32 && input_kind <= IMG_U8_HWC, "image_to_compute: bad argument");
    HIPT_CHECK_ARG(dst != nullptr && ((uintptr_t)dst & 255) == 0, "image_to_compute: null / unaligned destination");
    const void* img = nullptr;
    return image_to_compute(w, images, input_kind, lay, nseq, dst, &img, S(stream));
}

size_t hipt_vit256_range_workspace_bytes(const hipt_vit_weights* w, int nseq, int chunk) { return w && nseq > 0 ? vit256_range_bytes(w, nseq, chunk) : 0; }

int hipt_vit256_forward_range(const hipt_vit_weights* w, const void* images_cd, const hipt_image_layout* lay, int seq0, int nseq, int chunk,
                              float* out, void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(images_cd && lay && out && nseq > 0 && seq0 >= 0, "vit256_forward_range: null/empty argument");
    return vit256_range_impl(w, images_cd, lay, seq0, nseq, chunk, out, workspace, ws_bytes, S(stream));
}

// the same over fp32 pixels where the embedding kernel reads them itself (embed32.hip): no image in the compute dtype is needed
size_t hipt_vit256_range_px_workspace_bytes(const hipt_vit_weights* w, const hipt_image_layout* lay, int nseq, int chunk) {
    if (!w || !lay || nseq <= 0 || lay->patch_h <= 0 || lay->patch_w <= 0) return 0;
    const size_t slot = al256(hipt_embed32_packed_bytes() + 256);
    // (the pointer's alignment is checked at the call; any non-null 16-byte aligned value stands in for it here)
    return embed_fused_ok(w, (const void*)16, lay, slot) ? vit256_range_bytes(w, nseq, chunk) + slot : 0;
}

int hipt_vit256_forward_range_px(const hipt_vit_weights* w, const float* images, const hipt_image_layout* lay, int seq0, int nseq, int chunk,
                                 float* out, void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(images && lay && out && nseq > 0 && seq0 >= 0, "vit256_forward_range_px: null/empty argument");
    const size_t slot = al256(hipt_embed32_packed_bytes() + 256), nrange = vit256_range_bytes(w, nseq, chunk);
    if (!embed_fused_ok(w, images, lay, slot)) {
        hipt_set_error("vit256_forward_range_px: this model / layout has no pixel-reading embedding (hipt_vit256_range_px_workspace_bytes returns 0)");
        return HIPT_E_UNSUPPORTED;
    }
    if (ws_bytes < nrange + slot || ((uintptr_t)workspace & 255)) {
        hipt_set_error("vit256_forward_range_px: workspace %zu B too small / unaligned (need %zu)", ws_bytes, nrange + slot);
        return HIPT_E_WORKSPACE;
    }
    void* pk = (char*)workspace + nrange;
    if ((rc = hipt_embed32_pack_launch(w->embed_w, pk, S(stream)))) return rc;
    return vit256_range_impl(w, images, lay, seq0, nseq, chunk, out, workspace, nrange, S(stream), pk);
}

int hipt_vit4k_forward(const hipt_vit_weights* w, const float* tokens_in, int nseq, float* out, void* workspace,
                       size_t ws_bytes, void* stream) {
    int rc = check_vit(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(out && nseq > 0, "vit4k_forward: null/empty argument");
    hipStream_t st = S(stream);
    Carver c(workspace, ws_bytes);
    float* x = (float*)c.take((size_t)nseq * w->ntok * w->dim * 4);
    BlockScratch s = carve_blocks(c, w, nseq);
    void* tokT = c.take((size_t)nseq * w->ntok * w->embed_k * 2);
    if (!c.ok()) {
        hipt_set_error("vit4k_forward: workspace %zu B too small / unaligned (need %zu)", ws_bytes, c.used);
        return HIPT_E_WORKSPACE;
    }
    const void* tok = tokens_in;
    const int64_t n = (int64_t)nseq * (w->ntok - 1) * w->embed_k;
    if (w->dtype == HIPT_BF16 && n > 0) {
        if ((rc = hipt_f32_to_bf16_launch(tokens_in, tokT, n, st))) return rc;
        tok = tokT;
    }
    // ALL the regions of a call go through the small-call kernels together (round 6; `force_small`): one wave per 16 x 32 output tile, rows
    // independent bit for bit, the attention one workgroup per (region, head) -- which kernels a region's 257 rows meet, and the bits they
    // write, do not depend on how many regions share the call: one region alone, eight gathered by extract_slide and a ragged tail of three
    // agree exactly (tests).  Thirty launches per call whatever its size (round 5 walked the regions in groups of four: 30 launches per
    // group, linear in nseq); the phi GEMM takes the small kernel too, for the same reason.
    if ((rc = embed4k(w, tok, nseq, x, st, 1))) return rc;
    if ((rc = run_blocks(w, x, nseq, 0, w->depth, nullptr, s, st, false, nullptr, false, nullptr, false, true))) return rc;
    return hipt_layernorm_launch(x, (int64_t)w->ntok * w->dim, w->norm_w, w->norm_b, out, HIPT_F32, w->dim, nseq, w->dim, w->ln_eps, st);
}

static hipt_image_layout region_layout(int W, int H) {
    hipt_image_layout lay;
    lay.grid_w = W / 256;
    lay.grid_h = H / 256;
    lay.patch_h = lay.patch_w = 256;
    lay.row_stride = H;
    lay.chan_stride = (int64_t)W * H;
    lay.batch_stride = 3 * lay.chan_stride;
    return lay;
}

size_t hipt_hipt4k_workspace_bytes(const hipt_vit_weights* w256, const hipt_vit_weights* w4k, int nreg, int w_256, int h_256,
                                   int chunk) {
    const hipt_image_layout lay = region_layout(w_256 * 256, h_256 * 256);
    const int nseq = nreg * w_256 * h_256;
    return hipt_vit256_forward_workspace_bytes(w256, &lay, nseq, chunk) + hipt_vit4k_forward_workspace_bytes(w4k, nreg) +
           al256((size_t)nseq * w256->dim * 4);
}

static int hipt4k_forward_impl(const hipt_vit_weights* w256, const hipt_vit_weights* w4k, const void* regions, int kind, int nreg, int W,
                               int H, int chunk, float* cls256_out, float* out, void* workspace, size_t ws_bytes, void* stream) {
    HIPT_CHECK_ARG(w256 && w4k && regions && out && nreg > 0, "hipt4k_forward: null/empty argument");
    HIPT_CHECK_ARG(W > 0 && H > 0 && W % 256 == 0 && H % 256 == 0, "hipt4k_forward: region %dx%d must be cropped to multiples of 256",
                   W, H);
    const int per = (W / 256) * (H / 256), nseq = nreg * per;
    HIPT_CHECK_ARG(w4k->ntok == per + 1, "hipt4k_forward: ViT-4K weights prepared for %d tokens, region has %d", w4k->ntok, per + 1);
    HIPT_CHECK_ARG(w4k->embed_k == w256->dim, "hipt4k_forward: ViT-4K input width %d != ViT-256 width %d", w4k->embed_k, w256->dim);
    const hipt_image_layout lay = region_layout(W, H);
    const size_t n256 = hipt_vit256_forward_workspace_bytes(w256, &lay, nseq, chunk) + image_extra_bytes(w256, &lay, nseq, kind);
    const size_t n4k = hipt_vit4k_forward_workspace_bytes(w4k, nreg);
    const size_t ncls = al256((size_t)nseq * w256->dim * 4);
    if (ws_bytes < n256 + n4k + ncls || ((uintptr_t)workspace & 255)) {
        hipt_set_error("hipt4k_forward: workspace %zu B too small / unaligned (need %zu)", ws_bytes, n256 + n4k + ncls);
        return HIPT_E_WORKSPACE;
    }
    char* ws = (char*)workspace;
    // cls256 [nreg * per, 384] token-major = nreg sequences of `per` tokens: exactly phi's input (hipt_4k.py:72-74)
    float* cls = cls256_out ? cls256_out : (float*)ws;
    int rc = vit256_forward_impl(w256, regions, kind, &lay, nseq, chunk, cls, ws + ncls, n256, stream);
    if (rc) return rc;
    return hipt_vit4k_forward(w4k, cls, nreg, out, ws + ncls + n256, n4k, stream);
}

int hipt_hipt4k_forward(const hipt_vit_weights* w256, const hipt_vit_weights* w4k, const float* regions, int nreg, int W, int H,
                        int chunk, float* cls256_out, float* out, void* workspace, size_t ws_bytes, void* stream) {
    return hipt4k_forward_impl(w256, w4k, regions, IMG_F32, nreg, W, H, chunk, cls256_out, out, workspace, ws_bytes, stream);
}

size_t hipt_hipt4k_u8_workspace_bytes(const hipt_vit_weights* w256, const hipt_vit_weights* w4k, int nreg, int w_256, int h_256, int chunk) {
    if (!w256 || !w4k || nreg <= 0 || w_256 <= 0 || h_256 <= 0) return 0;
    const hipt_image_layout lay = region_layout(w_256 * 256, h_256 * 256);
    return hipt_hipt4k_workspace_bytes(w256, w4k, nreg, w_256, h_256, chunk) + image_extra_bytes(w256, &lay, nreg * w_256 * h_256, IMG_U8_CHW);
}

int hipt_hipt4k_forward_u8(const hipt_vit_weights* w256, const hipt_vit_weights* w4k, const uint8_t* regions, int interleaved, int nreg,
                           int W, int H, int chunk, float* cls256_out, float* out, void* workspace, size_t ws_bytes, void* stream) {
    return hipt4k_forward_impl(w256, w4k, regions, interleaved ? IMG_U8_HWC : IMG_U8_CHW, nreg, W, H, chunk, cls256_out, out, workspace,
                               ws_bytes, stream);
}

int hipt_u8_normalize(const void* src, int interleaved, int64_t n_images, int64_t plane, void* dst, int dst_dtype, void* stream) {
    return hipt_u8_normalize_launch(src, interleaved, n_images, plane, dst, dst_dtype, S(stream));
}

// ------------------------------------------------------------------------------------------------
// CLAM_SB / ABMIL
// ------------------------------------------------------------------------------------------------
static int check_clam(const hipt_clam_weights* w) {
    HIPT_CHECK_ARG(w != nullptr, "clam: null weights");
    HIPT_CHECK_ARG(w->dtype == HIPT_F32 || w->dtype == HIPT_BF16, "clam: bad dtype %d", w->dtype);
    HIPT_CHECK_ARG(w->s1 > 0 && w->s2 > 0, "clam: bad widths [%d,%d,%d]", w->s0, w->s1, w->s2);
    return HIPT_OK;
}

static size_t clam_partials_bytes(const hipt_clam_weights* w, int N) {
    const size_t g = 1024;  // fused: <= 512 workgroups; generic pool: <= 1024 row blocks
    return al256(g * (2 + w->s1) * 4);
}

// The ticket block is the FIRST 256 bytes of the workspace, whatever the model's widths, and nothing else ever writes there
// (one workspace may serve several CLAM modules of different widths / paths on a stream: the generic path's scratch must not
// run over the streaming kernels' arrival counter).
size_t hipt_clam_ticket_offset(const hipt_clam_weights* w, int N) { return 0; }

size_t hipt_clam_workspace_bytes(const hipt_clam_weights* w, int N) {
    // ticket | partials | gmax | h1 fp32 | ab fp32 | h1 in dtype (generic path)
    return 256 + clam_partials_bytes(w, N) + 256 + al256((size_t)N * w->s1 * 4) + al256((size_t)N * 2 * w->s2 * 4) +
           al256((size_t)N * w->s1 * 2);
}

size_t hipt_clam_stream_packed_bytes(const hipt_clam_weights* w) { return w ? hipt_clam_stream_image_bytes(w) : 0; }

int hipt_clam_stream_pack(const hipt_clam_weights* w, void* out, void* stream) {
    HIPT_CHECK_ARG(w != nullptr, "clam_stream_pack: null weights");
    return hipt_clam_stream_pack_launch(w, out, S(stream));
}

static int gated_scores(const hipt_clam_weights* w, const void* x, int xdtype, int N, float* ab, void* xT, float* A,
                        hipStream_t st) {
    // ab = x @ [Wa;Wb]^T + [ba;bb]  (x: [N,S1] in xdtype), then the gate
    const int kb = w->dtype == HIPT_F32 ? 32 : 64, n2 = 2 * w->s2;
    int rc;
    if (w->s1 % kb == 0 && n2 % 4 == 0) {
        const void* a = x;
        if (xdtype != w->dtype) {  // fp32 h1 -> bf16 operand
            if ((rc = hipt_f32_to_bf16_launch((const float*)x, xT, (int64_t)N * w->s1, st))) return rc;
            a = xT;
        }
        rc = linear(a, w->s1, w->wab, w->s1, w->bab, nullptr, ab, n2, N, n2, w->s1, w->dtype, HIPT_EPI_OUT_F32, st);
    } else {
        rc = hipt_small_ab_launch(x, xdtype, N, w->s1, n2, w->wab, w->dtype, w->bab, ab, st);
    }
    if (rc) return rc;
    return hipt_gate_launch(ab, n2, N, w->s2, w->wc, w->bc, A, st);
}

int hipt_clam_sb_forward(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* M,
                         float* logits, float* Y_prob, int64_t* Y_hat, void* workspace, size_t ws_bytes, void* stream) {
    int rc = check_clam(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(bag && A_raw && N > 0, "clam_sb_forward: null/empty bag (N=%d)", N);
    HIPT_CHECK_ARG(attention_only || (M && logits && Y_prob && Y_hat), "clam_sb_forward: null output");
    HIPT_CHECK_ARG(((uintptr_t)bag & 15) == 0, "clam_sb_forward: bag must be 16-byte aligned");
    const int kb = w->dtype == HIPT_F32 ? 32 : 64;
    if (w->s0 % kb != 0 || w->s1 % 4 != 0) {
        hipt_set_error("clam_sb_forward: S0=%d must be a multiple of %d and S1=%d of 4", w->s0, kb, w->s1);
        return HIPT_E_UNSUPPORTED;
    }
    if (ws_bytes < hipt_clam_workspace_bytes(w, N) || ((uintptr_t)workspace & 255)) {
        hipt_set_error("clam_sb_forward: workspace %zu B too small / unaligned (need %zu)", ws_bytes, hipt_clam_workspace_bytes(w, N));
        return HIPT_E_WORKSPACE;
    }
    hipStream_t st = S(stream);
    Carver c(workspace, ws_bytes);
    unsigned* ticket = (unsigned*)c.take(256);  // (hipt_clam_ticket_offset() = 0: zero before the first use, zero after every call)
    float* partials = (float*)c.take(clam_partials_bytes(w, N));
    float* gmax = (float*)c.take(256);
    int G = 0;
    if (hipt_clam_stream_supported(w)) {  // bf16 [S0,128,64]: weight-stationary streaming kernel
        PROF(PC_ABMIL, hipt_clam_stream_launch(w, bag, N, attention_only, A_raw, partials, &G, ticket, M, logits, Y_prob, Y_hat, st));
        if (!attention_only && G == 0) return HIPT_OK;
    } else if (hipt_clam_fused_supported(w)) {
        PROF(PC_ABMIL, hipt_clam_fused_launch(w, bag, N, attention_only, A_raw, partials, &G, st));
    } else {
        float* h1 = (float*)c.take((size_t)N * w->s1 * 4);
        float* ab = (float*)c.take((size_t)N * 2 * w->s2 * 4);
        void* h1T = c.take((size_t)N * w->s1 * 2);
        if ((rc = linear(bag, w->s0, w->w1, w->s0, w->b1, nullptr, h1, w->s1, N, w->s1, w->s0, w->dtype,
                         HIPT_EPI_RELU | HIPT_EPI_OUT_F32, st)))
            return rc;
        if ((rc = gated_scores(w, h1, HIPT_F32, N, ab, h1T, A_raw, st))) return rc;
        if (!attention_only && (rc = hipt_pool_launch(A_raw, h1, N, w->s1, gmax, partials, &G, st))) return rc;
    }
    if (attention_only) return HIPT_OK;
    PROF(PC_COMBINE, hipt_clam_combine_launch(partials, G, w, M, logits, Y_prob, Y_hat, st));
    return HIPT_OK;
}

int hipt_clam_mb_supported(const hipt_clam_weights* w) { return w && check_clam(w) == HIPT_OK && hipt_clam_mb_stream_supported(w) ? 1 : 0; }

size_t hipt_clam_mb_workspace_bytes(const hipt_clam_weights* w, int N) {
    // ticket | partials of <= 128 workgroups x K branches | h1 as a bf16 image
    return 256 + al256((size_t)128 * 4 * (4 + 128) * 4) + al256(hipt_clam_mb_h1_bytes(N > 0 ? N : 1));
}

int hipt_clam_mb_forward(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* M, float* logits, void* workspace,
                         size_t ws_bytes, void* stream) {
    int rc = check_clam(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(bag && A_raw && N > 0, "clam_mb_forward: null/empty bag (N=%d)", N);
    HIPT_CHECK_ARG(attention_only || (M && logits), "clam_mb_forward: null output");
    HIPT_CHECK_ARG(((uintptr_t)bag & 15) == 0, "clam_mb_forward: bag must be 16-byte aligned");
    if (!hipt_clam_mb_stream_supported(w)) {
        hipt_set_error("clam_mb_forward: no one-pass form for this configuration (bf16 [384|192,128,64], 2..4 branches = classes, stream_pk, bound < 60): "
                       "call hipt_clam_sb_forward per branch");
        return HIPT_E_UNSUPPORTED;
    }
    if (ws_bytes < hipt_clam_mb_workspace_bytes(w, N) || ((uintptr_t)workspace & 255)) {
        hipt_set_error("clam_mb_forward: workspace %zu B too small / unaligned (need %zu)", ws_bytes, hipt_clam_mb_workspace_bytes(w, N));
        return HIPT_E_WORKSPACE;
    }
    hipStream_t st = S(stream);
    Carver c(workspace, ws_bytes);
    unsigned* ticket = (unsigned*)c.take(256);
    float* partials = (float*)c.take((size_t)128 * 4 * (4 + 128) * 4);
    void* h1 = c.take(hipt_clam_mb_h1_bytes(N));
    // (the two launches are booked apart: 'abmil_fused' = the streaming pass, 'abmil_combine' = the pooling pass)
    PROF(PC_ABMIL, hipt_clam_mb_stream_launch(w, bag, N, 1, A_raw, h1, partials, ticket, M, logits, st));  // (passes = 1: the streaming pass)
    if (!attention_only) PROF(PC_COMBINE, hipt_clam_mb_stream_launch(w, bag, N, 2, A_raw, h1, partials, ticket, M, logits, st));  // (passes = 2: the pooling pass)
    return HIPT_OK;
}

int hipt_attn_net_gated(const hipt_clam_weights* w, const void* x, int N, float* A, void* workspace, size_t ws_bytes,
                        void* stream) {
    int rc = check_clam(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(x && A && N > 0, "attn_net_gated: null/empty input");
    const size_t need = al256((size_t)N * 2 * w->s2 * 4);
    if (ws_bytes < need || ((uintptr_t)workspace & 255)) {
        hipt_set_error("attn_net_gated: workspace %zu B too small / unaligned (need %zu)", ws_bytes, need);
        return HIPT_E_WORKSPACE;
    }
    return gated_scores(w, x, w->dtype, N, (float*)workspace, nullptr, A, S(stream));
}

int hipt_clam_gather_h1(const hipt_clam_weights* w, const void* bag, const int64_t* idx, int n_idx, float* out, void* stream) {
    int rc = check_clam(w);
    if (rc) return rc;
    HIPT_CHECK_ARG(bag && idx && out && n_idx > 0, "clam_gather_h1: null/empty argument");
    return hipt_gather_h1_launch(w, bag, idx, n_idx, out, S(stream));
}

}  // extern "C"
