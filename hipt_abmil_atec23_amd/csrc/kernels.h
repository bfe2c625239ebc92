// Internal (library-private) launch interface between capi.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hipt_abmil.h"

enum { ALOAD_PLAIN = 0, ALOAD_IM2COL = 1 };
enum { EPI_ROWMAP = 32 };  // internal epilogue flag: token rows -> [CLS]-skipping rows, + pos table

struct GemmParams {
    const void* A;
    int64_t lda;
    const void* W;
    int64_t ldw;
    int M, N, K;
    int rpt;  // rows per M tile = rows per sequence (<= 272); 0 -> plain 272-row tiles
    const float* bias;
    const float* resid;
    void* out;
    int64_t ldc;
    // EPI_ROWMAP
    const float* pos;
    int rows_per_seq;
    // ALOAD_IM2COL
    hipt_image_layout im;
    int im_nty, im_ntx, im_seq0;
    // LayerNorm prologue (small calls only, hipt_gemm_ln_supported): A is then the fp32 rows [M, K] whatever the compute dtype, and
    // the product is taken with LayerNorm(A) rounded to the compute dtype -- what misc.hip's ln_kernel + this GEMM give in two launches
    const float *ln_w, *ln_b;
    float ln_eps;
    // rows gathered from an activation image (small calls only, hipt_gemm_arows_supported): A is a bf16 image [.., 384] and GEMM row r
    // is image row r * a_row_step (the [CLS] rows of the sequences: a_row_step = tokens per sequence)
    int a_row_step;
    // small-M kernel only: k in ASCENDING 32-byte steps (no pairing, no staggered start) -- the k sets and the order of the tiled kernel's
    // MFMAs, so that an output element is the same bits whichever of the two kernels a call's row count selects (capi.hip: rows_linear)
    int asc;
    // the small-M kernels whatever M is (round 6): the second-level ViT runs ALL the regions of a call through them in one launch per operator --
    // they are row independent bit for bit, so a region's features do not depend on how many regions share the call (capi.hip: hipt_vit4k_forward)
    int small_any;
};
bool hipt_gemm_arows_supported(int M, int K, int dtype, int aload, int flags);

bool hipt_gemm_ln_supported(int M, int K, int aload, int flags, bool any_m = false);
int hipt_gemm_launch(const GemmParams& p, int dtype, int aload, int flags, hipStream_t st);

// Patch embedding of ViT-256 straight from the fp32 image (embed32.hip): x[seq, 1 + t, :] = Conv2d_k16_s16(pixels) + bias + pos[1 + t]
// for the 16 x 16 tokens of every 256 x 256 patch; bf16 MFMAs on pixels rounded to bf16 in registers (no bf16 copy of the image).
struct EmbedParams {
    const void* img;         // image tensor, addressed through `im` (include/hipt_abmil.h, hipt_image_layout)
    int kind;                // 0: fp32 [.., 3, W, H]; 1: uint8 in the same layout; 2: uint8 interleaved [.., W, H, 3] (`im` still gives W, H as strides)
    hipt_image_layout im;
    int nty, ntx;            // tokens per patch along dim2 / dim3 (ntx == 16, nty % 8 == 0)
    int seq0, nseq;          // sequences (patches) [seq0, seq0 + nseq) of the tensor
    const void* wpk;         // the Conv2d weight [D, 768] as the kernel's fragment image (hipt_embed32_pack_launch)
    const float* bias;       // [D]
    const float* pos;        // [ntok, D] (row 0 = [CLS])
    float* x;                // [nseq, ntok, D] fp32; row 0 of every sequence is left alone
    int ntok;
    int* counter;            // device int the launcher zeroes on the stream: the kernel's tile queue
    int ntiles;              // (set by the launcher)
    // optional: xn_out != null -> x is written as the fp32 ACTIVATION IMAGE (below) and LayerNorm(x; ln_w, ln_b, ln_eps) -- the first block's
    // LayerNorm-1 -- as the bf16 image xn_out (nseq * ntok % 16 == 0); the [CLS] rows of both: hipt_cls_init_img_launch
    void* xn_out;
    const float* ln_w;
    const float* ln_b;
    float ln_eps;
};
bool hipt_embed32_supported(int dtype, int D, int K, int nty, int ntx);
size_t hipt_embed32_packed_bytes();
int hipt_embed32_pack_launch(const void* w, void* packed, hipStream_t st);
int hipt_embed32_launch(const EmbedParams& p, hipStream_t st);

// A-stationary GEMM (seqgemm.hip): one workgroup per sequence, activations in registers, optional fused LayerNorm
// "Activation images" (pipelined D = 384 kernels, M % 16 == 0): a [M, 384] activation stored fragment by fragment in
// the order the MFMA operand / accumulator layout wants it, so that every load / store instruction of a wave covers 1 KiB
// of consecutive bytes.  Fragment F = rows [16F, 16F+16); lane = 16 g + li holds row li, 16-byte chunk (g + 4c):
//   bf16: element offset  F * 6144 + c * 512 + lane * 8                  (c = 0..11: columns (g + 4c) * 8 .. + 7)
//   fp32: float offset    F * 6144 + c * 512 + h * 256 + lane * 4        (h = 0/1: columns (g + 4c) * 8 + 4h .. + 3)
// A fragment occupies the same bytes as its 16 rows do row-major, so a kernel may convert in place fragment by fragment.
struct SeqGemmParams {
    const void* A;       // LN: fp32 x rows; else bf16 rows
    int64_t lda;         // elements
    const float* ln_w;
    const float* ln_b;
    float ln_eps;
    const void* W;       // bf16 [N, K]
    const void* wpk;     // W pre-packed in ring order (hipt_seqgemm_pack_launch): what selects the pipelined kernel (null: the generic one)
    int img;             // (pipelined kernel only) activation images: bit 0 = A (bf16, no LayerNorm), bit 1 = out (N = 384);
                         // bit 2 = out (N = 1152) head-major [sequence][q/k/v][head][token][64], out_ntok tokens per sequence
    int out_ntok;
    int M, N, K;
    unsigned long long* stamps;  // debug: per-workgroup phase timestamps (100 MHz), or null
    int full_tiles;      // (set by the launcher) row tiles run whole; the rest are split nsplit ways over N
    int nsplit;
    const float* bias;
    void* out;           // bf16 [M, ldc]
    int64_t ldc;
    int* counter;        // (pipelined kernel) device int the launcher zeroes on the stream: the tile queue
    int counter_zeroed;  // 1: the caller guarantees *counter == 0 at launch (seqgemm_pipe.hip skips its memset); the kernel leaves it 0 again
    int ntiles;          // (set by the launcher)
};
bool hipt_seqgemm_supported(int dtype, int K);
int hipt_seqgemm_launch(const SeqGemmParams& p, bool ln, int flags, hipStream_t st);  // dispatches to the pipelined kernel when it applies
bool hipt_seqgemm_pipe_supported(int dtype, int K, int N, bool ln, int flags);
int hipt_seqgemm_pipe_launch(const SeqGemmParams& p, bool ln, hipStream_t st);
// W as ONE contiguous image (N * K bf16, re-ordered): ring unit (N tile nt, k half kh) at byte (2 nt + kh) * 48 KiB,
// byte for byte what its LDS slot holds -- a DMA piece reads 1 KiB of consecutive bytes (see hipt_mlp_pack_launch).
int hipt_seqgemm_pack_launch(const void* W, int N, int K, void* packed, hipStream_t st);

// Fused MLP sub-block (mlp.hip): x <- x + y1 + fc2(GELU(fc1(LN2(x + y1))))
struct MlpParams {
    float* x;            // fp32 [M, D] residual stream, updated in place
    const void* y1;      // bf16 [M, D] attention-branch output still to be added (or null); fold: the ATTENTION OUTPUT, before proj
    const float* bproj;  // fold: the proj bias [D]
    int fold;            // 1 (mlp16.hip, image format 3 only): the output projection runs at the head of the tile, no proj launch before this kernel
    const float* ln_w;
    const float* ln_b;
    float ln_eps;
    const void* w1;      // bf16 [hidden, D]
    const float* b1;
    const void* w2;      // bf16 [D, hidden]
    const float* b2;
    int M, D, hidden;
    const void* wpk;     // optional (streaming kernel only): both weights pre-packed in ring order (hipt_mlp16_pack_launch)
    int wpk_fmt;         // format of wpk: 2 = mlp16.hip's fragment image, 3 = the same behind six proj units (1 was the retired 32x32x16 form's)
    int img;             // (pipelined kernel only) activation images: bit 0 = y1, xn_out and the updated x, bit 1 = x on entry
    int* counter;        // device int the launcher zeroes on the stream: the kernel's tile queue
    int counter_zeroed;  // 1: the caller guarantees *counter == 0 at launch (mlp16.hip skips its memset); these kernels leave it 0 again
    // optional (pipelined kernel only): LayerNorm-1 of the NEXT block applied to the updated rows, written as bf16
    // [M, D] -- the next block's QKV GEMM then loads operands directly instead of fp32 rows + LayerNorm
    const float* ln_next_w;
    const float* ln_next_b;
    void* xn_out;
    int full_tiles;      // (set by the launcher) 128-row tiles; tiles beyond are 16-row tail tiles
    int ntiles;          // (set by the launcher)
    int stagger;         // (set by the launcher) start offset between workgroup groups, 10 ns ticks (0 = none)
    unsigned long long* stamps;
};
bool hipt_mlp_supported(int dtype, int D, int hidden);
int hipt_mlp_launch(const MlpParams& p, hipStream_t st);  // dispatches to the packed-weight D = 384 kernel (mlp16.hip) when it applies
// The streaming kernel (mlp16.hip, 16x16x32 MFMAs): its weight stream is ONE contiguous image (2 * hidden * D bf16 = the two
// matrices, re-ordered; format 2): unit after unit in the order a tile pass consumes them, each unit byte for byte what its LDS ring
// slot holds, so that a DMA piece reads 1 KiB of consecutive bytes instead of 8 row segments of 128 B (2.4x the L2 -> LDS rate).
// (Its 32x32x16 twin of rounds 2-4 -- format 1 -- lives in tools/experiments/mlp32_r4.hip: DESIGN.md.)
bool hipt_mlp16_supported(int dtype, int D, int hidden);
int hipt_mlp16_launch(const MlpParams& p, hipStream_t st);
size_t hipt_mlp16_packed_bytes(int D, int hidden, bool with_proj);
int hipt_mlp16_pack_launch(const void* w1, const void* w2, int D, int hidden, void* packed, hipStream_t st, const void* wproj = nullptr);

int hipt_layernorm_launch(const float* x, int64_t x_stride, const float* w, const float* b, void* out, int out_dtype,
                          int64_t out_stride, int rows, int D, float eps, hipStream_t st);

int hipt_attention_launch(const void* qkv, void* out, float* probs, int B, int ntok, int heads, int dh, float scale,
                          int dtype, hipStream_t st, int out_img = 0, int qkv_hm = 0);  // dispatches to the bf16 / head-dim-64 kernel when it applies
bool hipt_attention64_supported(int dtype, int dh, int ntok, bool want_probs);
int hipt_attention64_launch(const void* qkv, void* out, int B, int ntok, int heads, float scale, hipStream_t st, int out_img = 0, int qkv_hm = 0);

// QKV projection + attention of a LayerNorm-chained ViT-256 block in one kernel (qkv_attention.hip): xn_img = LayerNorm-1(x) and
// out_img = the attention output, both bf16 activation images [nseq * 257, 384]; qkv_cls = bf16 [nseq, 1152] (+ 1 KiB readable
// slack) = q | k | v of the [CLS] rows (a side GEMM); wpk = qkv_w in the kernel's operand order (hipt_qkv_attn_pack_launch)
bool hipt_qkv_attn_supported(int dtype, int D, int heads, int ntok);
size_t hipt_qkv_attn_packed_bytes();
int hipt_qkv_attn_pack_launch(const void* qkv_w, void* packed, hipStream_t st);
int hipt_qkv_attn_launch(const void* xn_img, const void* wpk, const float* qkv_b, const void* qkv_cls, void* out_img, int nseq, float scale, hipStream_t st);
int hipt_qkv_attn_cls_launch(const void* xn_img, const void* wpk, const float* qkv_b, const void* qkv_cls, void* out_rows, int nseq, float scale, hipStream_t st);

// attention of the [CLS] query only (bf16, head dim 64): out[B, heads*64] bf16 and/or probs[B, heads, ntok] fp32 (either may be null)
int hipt_attn_cls_launch(const void* qkv, void* out, float* probs, int B, int ntok, int heads, int dh, float scale, hipStream_t st);
// probabilities of the [CLS] query only, any compute dtype, head dim 32 / 64: probs[B, heads, ntok] fp32
int hipt_attn_cls_probs_launch(const void* qkv, float* probs, int B, int ntok, int heads, int dh, float scale, int dtype, hipStream_t st);
// dst[s, :] = src[s * seq_stride ...] : the first row of every sequence (fp32)
int hipt_gather_cls_launch(const float* src, float* dst, int nseq, int64_t seq_stride, int D, hipStream_t st, int img = 0);
int hipt_gather_cls_bf16_launch(const void* src, void* dst, int nseq, int ntok, int D, hipStream_t st, int img = 0);  // bf16 rows s * ntok of [.., 384]
// x[s, 0, :] = cls + pos[0]  for s in [0, nseq)
int hipt_cls_init_launch(float* x, const float* cls, const float* pos, int nseq, int ntok, int D, hipStream_t st);
// the same into ACTIVATION IMAGES (D = 384): row s * ntok of the fp32 image x_img = cls + pos[0], and of the bf16 image xn_img its LayerNorm
int hipt_cls_init_img_launch(float* x_img, void* xn_img, const float* cls, const float* pos, const float* ln_w, const float* ln_b, float ln_eps, int nseq,
                             int ntok, int D, hipStream_t st);
// out[i] = src[i] (+ (float)y[i] if y)  (fp32, bf16 branch, n % 8 == 0): lands the residual stream in the caller's buffer
int hipt_add_bf16_launch(float* out, const float* src, const void* y, int64_t n, hipStream_t st);
// uint8 image -> normalised compute-dtype image [n, 3, plane] (ToTensor + Normalize(0.5, 0.5)); hwc: src is [n, plane, 3]
int hipt_u8_normalize_launch(const void* src, int hwc, int64_t nimg, int64_t plane, void* dst, int dst_dtype, hipStream_t st);
// fp32 -> bf16 elementwise (n % 8 == 0)
int hipt_f32_to_bf16_launch(const float* in, void* out, int64_t n, hipStream_t st);

// ---- CLAM / ABMIL ----
bool hipt_clam_fused_supported(const hipt_clam_weights* w);
int hipt_clam_fused_launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw,
                           float* partials, int* n_partials, hipStream_t st);
bool hipt_clam_stream_supported(const hipt_clam_weights* w);
size_t hipt_clam_stream_image_bytes(const hipt_clam_weights* w);
int hipt_clam_stream_pack_launch(const hipt_clam_weights* w, void* out, hipStream_t st);
int hipt_clam_stream_launch(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw,
                            float* partials, int* n_partials, unsigned* ticket, float* M, float* logits, float* Y_prob,
                            int64_t* Y_hat, hipStream_t st);  // n_partials = 0: combine already done in the kernel
// CLAM_MB in one pass over the bag (abmil32.hip): `passes` bit 0 = the streaming kernel with w->n_att branches (A_raw, h1 as a bf16 image),
// bit 1 = the pooling kernel (M, logits)
bool hipt_clam_mb_stream_supported(const hipt_clam_weights* w);
size_t hipt_clam_mb_h1_bytes(int N);
int hipt_clam_mb_stream_launch(const hipt_clam_weights* w, const void* bag, int N, int passes, float* A_raw, void* h1_img, float* partials,
                               unsigned* ticket, float* M, float* logits, hipStream_t st);
int hipt_clam_combine_launch(const float* partials, int G, const hipt_clam_weights* w, float* M, float* logits,
                             float* Y_prob, int64_t* Y_hat, hipStream_t st);
int hipt_gate_launch(const float* ab, int64_t ld, int N, int S2, const float* wc, const float* bc, float* A,
                     hipStream_t st);
int hipt_small_ab_launch(const void* x, int xdtype, int N, int S1, int S2x2, const void* wab, int wdtype,
                         const float* bab, float* ab, hipStream_t st);
int hipt_pool_launch(const float* A, const float* h1, int N, int S1, float* gmax, float* partials, int* n_partials,
                     hipStream_t st);
int hipt_gather_h1_launch(const hipt_clam_weights* w, const void* bag, const int64_t* idx, int n_idx, float* out,
                          hipStream_t st);
