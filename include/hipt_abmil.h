/*
 * hipt_abmil.h — C ABI of libhipt_abmil.so (gfx950 / MI355X).
 *
 * The reference (scjjb/HIPT_ABMIL_ATEC23) is pure Python on torch.nn; it has no native
 * boundary of its own.  This header is the boundary the build introduces UNDER the
 * reference's Python call surface: each entry point replaces the
 * stock-PyTorch op sequence of one reference function, cited per declaration as
 * file:line relative to the reference checkout.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch-allocated); the
 *     library allocates nothing persistent and keeps no state between calls;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and returns immediately: no host synchronisation, no
 *     allocation, so calls may be captured into a hipGraph;
 *   - return value: HIPT_OK (0), or a negative HIPT_E_* code; the library never throws
 *     or aborts.  hipt_last_error() returns a static message for the calling thread;
 *   - `dtype` selects the arithmetic type of the GEMM operands: HIPT_F32 (exact fp32 MFMA,
 *     the reference's numerics, parity 1e-4) or HIPT_BF16 (bf16 operands, fp32 accumulate).
 *     LayerNorm statistics, softmax, biases, the residual stream and all outputs are fp32
 *     in both modes;
 *   - matrices are row-major; Linear weights keep torch's [out_features, in_features] layout.
 */
#ifndef HIPT_ABMIL_H
#define HIPT_ABMIL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 6): the HIPT_PACK_MLP image of format 3 is 2*D*hidden*2 + D*D*2 bytes (six proj units in front: ask
 * hipt_vit_packed_bytes, never compute the size) and packing it reads blocks[i].proj_w, which must be set first;
 * hipt_vit_mlp_unit added; formats 1 / HIPT_MLP32 are gone. */
#define HIPT_ABI_VERSION 5

enum { HIPT_F32 = 0, HIPT_BF16 = 1 };

enum {
    HIPT_OK = 0,
    HIPT_E_BADARG = -1,      /* shape / dtype / alignment the kernels do not support */
    HIPT_E_WORKSPACE = -2,   /* workspace too small */
    HIPT_E_LAUNCH = -3,      /* hipLaunch / hip runtime error (see hipt_last_error) */
    HIPT_E_UNSUPPORTED = -4  /* valid request outside the implemented envelope */
};

int hipt_abi_version(void);
const char* hipt_last_error(void);

/* Optional per-kernel timing for benchmarking (not part of the reference surface): when enabled,
 * every kernel launch of the calls below is bracketed by HIP events on the caller's stream.
 * hipt_profile_read() synchronises on them, returns total milliseconds and launch counts per
 * category (hipt_profile_categories() entries, named by hipt_profile_category_name) and resets.
 * Returns HIPT_E_WORKSPACE if more launches were issued than the event pool (8192) holds. */
int hipt_profile_enable(int on);
int hipt_profile_categories(void);
const char* hipt_profile_category_name(int i);
int hipt_profile_read(float* ms, int* counts);

/* ------------------------------------------------------------------------------------
 * Transformer block weights: Block / Attention / Mlp
 * (HIPT_4K/vision_transformer.py:88-152, duplicated HIPT_4K/vision_transformer4k.py:94-158).
 * GEMM matrices are in `dtype`; LayerNorm affine terms and all biases are fp32.
 * ---------------------------------------------------------------------------------- */
typedef struct hipt_block_weights {
    const float* ln1_w;  const float* ln1_b;    /* blocks.i.norm1.{weight,bias}      [D]     */
    const void*  qkv_w;  const float* qkv_b;    /* blocks.i.attn.qkv.{weight,bias}   [3D,D]  */
    const void*  proj_w; const float* proj_b;   /* blocks.i.attn.proj.{weight,bias}  [D,D]   */
    const float* ln2_w;  const float* ln2_b;    /* blocks.i.norm2.{weight,bias}      [D]     */
    const void*  fc1_w;  const float* fc1_b;    /* blocks.i.mlp.fc1.{weight,bias}    [Dh,D]  */
    const void*  fc2_w;  const float* fc2_b;    /* blocks.i.mlp.fc2.{weight,bias}    [D,Dh]  */
    /* Optional (NULL = absent) pre-packed images of the bf16 GEMM matrices, written by hipt_vit_pack_weights from the
     * matrices above: the same values in the byte order of the streaming kernels' LDS ring, so that a 1 KiB DMA piece is
     * one run of consecutive bytes instead of eight 128-byte row segments (2.4x the L2->LDS rate on MI355X).  They
     * are a cache of qkv_w / proj_w / fc1_w+fc2_w: re-pack after the weights change. */
    const void*  qkv_pk;  const void* proj_pk;  const void* mlp_pk;
    /* The image format of mlp_pk: the value hipt_vit_mlp_pack_format returned when it was packed (3 = the streaming kernel's
     * fragment image on 16x16x32 MFMAs behind six units of proj_w: what this version packs; 2 = the same without the proj
     * units, still run; 0 = no packed form for this shape). */
    int32_t      mlp_pk_fmt;  int32_t reserved;
    /* Optional: qkv_w once more, head by head in the operand order of the fused QKV + attention kernel (HIPT_PACK_QKV_ATT;
     * ViT-256 shape only: D = 384, 6 heads of 64, 257 tokens).  NULL: LayerNorm-chained blocks run the QKV GEMM and the
     * attention as two kernels with the q|k|v tensor in HBM between them. */
    const void*  qkv_att_pk;
} hipt_block_weights;

#define HIPT_PACK_QKV  0
#define HIPT_PACK_PROJ 1
#define HIPT_PACK_MLP  2   /* fc1 and fc2 in one image, in the format hipt_vit_mlp_pack_format names */
#define HIPT_PACK_QKV_ATT 3

/* One ViT (ViT-256 `vit_small` or ViT-4K `vit4k_xs`, or any width the classes are built with).
 * `pos` is the ALREADY INTERPOLATED positional table for this token grid
 * (interpolate_pos_encoding, vision_transformer.py:213-233): input independent, computed once
 * on the host side with the reference's exact F.interpolate call and cached. */
typedef struct hipt_vit_weights {
    int32_t dtype;        /* HIPT_F32 | HIPT_BF16 (type of embed_w and of the block GEMM matrices) */
    int32_t dim;          /* D: 384 (ViT-256) / 192 (ViT-4K); multiple of 32                        */
    int32_t depth;        /* number of blocks                                                     */
    int32_t heads;        /* D / heads must be 32 or 64                                           */
    int32_t hidden;       /* MLP hidden width (4*D)                                               */
    int32_t ntok;         /* tokens per sequence incl. [CLS] (257); <= 288                        */
    int32_t embed_k;      /* K of the embedding GEMM: 3*16*16 = 768 (ViT-256) / 384 (ViT-4K phi)   */
    float   ln_eps;       /* eps of every LayerNorm (1e-6 for vit_small / vit4k_xs)                 */
    float   attn_scale;   /* Attention.scale = qk_scale or head_dim ** -0.5 (vision_transformer.py:112);
                             <= 0 selects head_dim ** -0.5                                          */
    int32_t reserved;
    const void*  embed_w; /* patch_embed.proj.weight viewed [D, 768]  /  phi.0.weight [D, 384]     */
    const float* embed_b; /* patch_embed.proj.bias / phi.0.bias                                   */
    const float* cls;     /* cls_token [D]                                                        */
    const float* pos;     /* interpolated pos_embed [ntok, D]                                     */
    const float* norm_w;  const float* norm_b;     /* final norm                                  */
    const hipt_block_weights* blocks;              /* HOST array of `depth` entries               */
} hipt_vit_weights;

/* Where the 16x16-pixel tokens of sequence b live inside the input image tensor
 * (PatchEmbed Conv2d k16 s16, vision_transformer.py:155-170, fused with the
 * unfold/rearrange patchify of hipt_4k.py:64-65):
 *   pixel(b, c, y, x) = img[ (b / (grid_w*grid_h)) * batch_stride + c * chan_stride
 *                            + (((b / grid_h) % grid_w) * patch_h + y) * row_stride
 *                            + (b % grid_h) * patch_w + x ]
 * A plain batch [B,3,w,h]: grid 1x1, patch_h=w, patch_w=h, row_stride=h, chan_stride=w*h,
 * batch_stride=3*w*h.  A region [1,3,W,H] cut into 256x256 patches: grid (W/256)x(H/256),
 * patch 256x256, row_stride=H, chan_stride=W*H. */
typedef struct hipt_image_layout {
    int32_t grid_w, grid_h;      /* patch grid per batch item (w_256, h_256)       */
    int32_t patch_h, patch_w;    /* pixels per patch along dim2 / dim3 (multiples of 16) */
    int64_t row_stride, chan_stride, batch_stride; /* in elements                  */
} hipt_image_layout;

/* ---- fine-grained operators (also the units the parity tests exercise) ---- */

/* nn.LayerNorm(D, eps) over rows (vision_transformer.py:138,142,195).  x fp32 rows of D with
 * stride x_stride; out rows in out_dtype with stride out_stride.  D % 64 == 0, D <= 2048. */
int hipt_layernorm(const float* x, int64_t x_stride, const float* w, const float* b,
                   void* out, int out_dtype, int64_t out_stride, int rows, int D, float eps,
                   void* stream);

/* out = epilogue(A[M,K] @ W[N,K]^T + bias)  — nn.Linear (vision_transformer.py:93-95,114,116).
 * flags: bit0 GELU(erf) after bias, bit1 add fp32 residual `resid` (ld = ldc), bit2 output fp32
 * (else `dtype`), bit4 ReLU after bias.  A and W are `dtype`; K % (HIPT_F32 ? 32 : 64) == 0,
 * N % 4 == 0, 16-byte aligned rows. */
enum { HIPT_EPI_GELU = 1, HIPT_EPI_RESID = 2, HIPT_EPI_OUT_F32 = 4, HIPT_EPI_RELU = 16 };
int hipt_linear(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                const float* resid, void* out, int64_t ldc, int M, int N, int K, int dtype,
                int flags, void* stream);

/* Attention core of Attention.forward (vision_transformer.py:119-128): for each (b, head)
 * softmax(q k^T * scale) v with q,k,v read in place from the qkv projection output
 * qkv[B, ntok, 3, heads, dh] (`dtype`); out[B, ntok, heads*dh] (`dtype`).  If `probs` is not
 * NULL the [B, heads, ntok, ntok] fp32 softmax probabilities are also written (the tensor
 * Block.forward(return_attention=True) returns, :148-149).  dh in {32, 64}; ntok <= 288. */
int hipt_attention(const void* qkv, void* out, float* probs, int B, int ntok, int heads, int dh,
                   float scale, int dtype, void* stream);

/* ---- ViT forward (VisionTransformer.forward :248-253 / VisionTransformer4K.forward :241-246) ---- */

/* Bytes of scratch hipt_vit*_prepare_tokens / hipt_vit_blocks need for `nseq` sequences processed at
 * once (hipt_vit256_prepare_tokens additionally needs the bf16 copy of the image tensor in
 * HIPT_BF16 mode: + 2 bytes per image element it addresses). */
size_t hipt_vit_workspace_bytes(const hipt_vit_weights* w, int nseq);

/* Pre-packed weight images (hipt_block_weights.*_pk).  hipt_vit_packed_bytes: size of the image of matrix `what`
 * (HIPT_PACK_*) of one block, 0 when this dtype / shape has no packed form (then leave the pointer NULL).
 * hipt_vit_pack_weights: write the image of block `block` into `out` (device memory of that size) from w->blocks[block]
 * (HIPT_PACK_MLP: in the format w->blocks[block].mlp_pk_fmt names -- set it first, from hipt_vit_mlp_pack_format).
 * Done once per set of weights by the module that owns them (vision_transformer.py:_PackedVit here). */
size_t hipt_vit_packed_bytes(const hipt_vit_weights* w, int what);
/* The fused-MLP image format hipt_vit_pack_weights(.., HIPT_PACK_MLP, ..) writes for this model, to be stored in
 * hipt_block_weights.mlp_pk_fmt beside the pointer: 3 = the streaming kernel's fragment image on 16x16x32 MFMAs with six
 * units of the block's proj matrix in front (the attention block's output projection then runs at the head of the fused MLP's
 * tiles: LayerNorm-chained blocks have no proj launch and no y1 tensor; hipt_vit_packed_bytes(HIPT_PACK_MLP) includes them),
 * 0 = this dtype / shape has no packed form (the generic kernel reads the row-major matrices).  An image packed as format 2
 * (the same without the proj units) by an older binding still runs, with proj as its own kernel.  (1 was the 32x32x16 form
 * of ABI versions before round 5: an image in that format is no longer run -- its model takes the generic kernels.)
 * The BYTES of an image are private to the library build that wrote them (since late round 6 the fused-MLP image holds W1 / 8 and
 * 8 * W2, exact power-of-two scalings its kernel undoes): pack with the library that will run the image, never carry one across builds. */
int hipt_vit_mlp_pack_format(const hipt_vit_weights* w);
int hipt_vit_pack_weights(const hipt_vit_weights* w, int block, int what, void* out, void* stream);
/* Scratch of the whole-forward calls below (residual stream + block scratch + bf16 input copy). */
size_t hipt_vit256_forward_workspace_bytes(const hipt_vit_weights* w, const hipt_image_layout* lay,
                                           int nseq, int chunk);
size_t hipt_vit4k_forward_workspace_bytes(const hipt_vit_weights* w, int nseq);

/* prepare_tokens for ViT-256 (vision_transformer.py:235-246): patchify + Conv2d(3,D,16,16) +
 * [CLS] + positional table.  images: fp32, addressed through `lay`; x: fp32 [nseq, ntok, D]. */
int hipt_vit256_prepare_tokens(const hipt_vit_weights* w, const float* images,
                               const hipt_image_layout* lay, int seq0, int nseq, float* x,
                               void* workspace, size_t ws_bytes, void* stream);

/* prepare_tokens for ViT-4K (vision_transformer4k.py:223-239): phi = Linear(384,D)+GELU, [CLS],
 * positional table.  tokens_in: fp32 [nseq, ntok-1, embed_k] token-major
 * (= x.flatten(2,3).transpose(1,2)); x: fp32 [nseq, ntok, D]. */
int hipt_vit4k_prepare_tokens(const hipt_vit_weights* w, const float* tokens_in, int nseq, float* x,
                              void* workspace, size_t ws_bytes, void* stream);

/* Run blocks [blk_begin, blk_end) in place on x (Block.forward :146-152).  If probs != NULL the
 * last block of the range stops after its attention and only writes the probabilities
 * (get_last_selfattention :255-262); x then holds the input of that block. */
int hipt_vit_blocks(const hipt_vit_weights* w, float* x, int nseq, int blk_begin, int blk_end,
                    float* probs, void* workspace, size_t ws_bytes, void* stream);

/* The attention unit of ONE LayerNorm-chained ViT-256 block as the streaming path runs it (Attention.forward before proj,
 * vision_transformer.py:121-128), exposed for the parity tests: xn_img = LayerNorm-1(x) and out_img = the attention output,
 * both bf16 "activation images" of [nseq * 257, 384] (16-row fragments in MFMA operand order: element (row 16 F + i,
 * column 32 c + 8 g + e) at F * 6144 + c * 512 + (16 g + i) * 8 + e).  fused != 0: the QKV projection runs inside the
 * attention kernel (needs blocks[block].qkv_att_pk); fused == 0: QKV GEMM + attention kernel with q | k | v in the workspace
 * (needs blocks[block].qkv_pk).  bf16 ViT-256 shape only (D = 384, 6 heads, 257 tokens), nseq * 257 a multiple of 16.
 * workspace >= hipt_vit_workspace_bytes(w, nseq); out_img must not alias xn_img. */
int hipt_vit_attention_unit(const hipt_vit_weights* w, int block, const void* xn_img, int nseq, void* out_img, int fused,
                            void* workspace, size_t ws_bytes, void* stream);

/* The other half of a LayerNorm-chained ViT-256 block as the streaming path runs it, for per-kernel benches and parity tests:
 * x <- x + proj(att) + b_proj;  x <- x + fc2(GELU(fc1(LayerNorm-2(x))))  (Block.forward, vision_transformer.py:146-152 with Mlp.forward
 * :98-104) in ONE launch of the fused proj + MLP kernel.  x_img: the fp32 residual stream [nseq * 257, 384] as an activation image,
 * updated in place; att_img: the attention output (before proj) as a bf16 activation image (hipt_vit_attention_unit's out_img);
 * xn_out_img: NULL, or where LayerNorm-1 of block + 1 (of `block` itself for the last one) of the updated rows goes as a bf16 image
 * (may alias att_img: a workgroup loads all attention rows of its tile before it writes them).  Needs blocks[block].mlp_pk in format 3
 * (bf16, D = 384, hidden % 128 == 0), nseq * 257 a multiple of 16; workspace >= 256 bytes (the kernel's tile queue). */
int hipt_vit_mlp_unit(const hipt_vit_weights* w, int block, float* x_img, const void* att_img, int nseq, void* xn_out_img,
                      void* workspace, size_t ws_bytes, void* stream);

/* [CLS] row of the last block's attention map: probs_cls[nseq, heads, ntok] fp32 =
 * get_last_selfattention(x)[:, :, 0, :] (vision_transformer.py:255-262 as consumed by the heat-maps,
 * HIPT_4K/hipt_4k.py:143-158) without materialising [nseq, heads, ntok, ntok].  x = prepared tokens, modified.
 * Both compute dtypes, head dim 32 (ViT-4K) and 64 (ViT-256).  workspace >= hipt_vit_workspace_bytes(). */
int hipt_vit_cls_attention(const hipt_vit_weights* w, float* x, int nseq, float* probs_cls,
                           void* workspace, size_t ws_bytes, void* stream);

/* Final LayerNorm; cls_only=1 -> out[nseq, D] = norm(x)[:,0] (:252-253), else out[nseq, ntok, D]
 * (get_intermediate_layers :264-272). */
int hipt_vit_head(const hipt_vit_weights* w, const float* x, int nseq, int cls_only, float* out,
                  void* stream);

/* Whole ViT-256 forward over nseq patches in chunks of `chunk` sequences (chunk <= 0: library
 * default) -> out[nseq, D] fp32. */
int hipt_vit256_forward(const hipt_vit_weights* w, const float* images, const hipt_image_layout* lay,
                        int nseq, int chunk, float* out, void* workspace, size_t ws_bytes, void* stream);

/* The same in two steps, for callers that spread the patches of ONE call over several HIP streams (a batch-of-one region on
 * two streams: each stream's kernels fill the CUs the other's leave idle in their ragged last round of tiles):
 *   hipt_image_to_compute  the input tensor in the compute dtype (input_kind 0: fp32 [..,3,W,H]; 1: uint8 [..,3,W,H]; 2: uint8
 *                          interleaved [..,W,H,3], both normalised (x / 255 - 0.5) / 0.5) into dst
 *                          (hipt_image_compute_bytes(); 0 = fp32 input in fp32 mode: use the input where it lies);
 *   hipt_vit256_forward_range  ViT-256 over sequences [seq0, seq0 + nseq) of that tensor -> out[nseq, D]
 *                          (workspace >= hipt_vit256_range_workspace_bytes(w, nseq, chunk)). */
size_t hipt_image_compute_bytes(const hipt_vit_weights* w, const hipt_image_layout* lay, int nseq, int input_kind);
int hipt_image_to_compute(const hipt_vit_weights* w, const void* images, int input_kind, const hipt_image_layout* lay,
                          int nseq, void* dst, void* stream);
size_t hipt_vit256_range_workspace_bytes(const hipt_vit_weights* w, int nseq, int chunk);
int hipt_vit256_forward_range(const hipt_vit_weights* w, const void* images_cd, const hipt_image_layout* lay,
                              int seq0, int nseq, int chunk, float* out, void* workspace, size_t ws_bytes, void* stream);
/* The same straight from fp32 pixels, for models / layouts whose patch embedding reads them itself (bf16 ViT-256 on 256 x 256
 * patches): no compute-dtype copy of the image.  hipt_vit256_range_px_workspace_bytes returns 0 where that is not available
 * (then: hipt_image_to_compute + hipt_vit256_forward_range).
 * Bits: the pixel-reading embedding hands the first block LayerNorm-ed operands itself (bf16, whole 16-row fragments), which the embedding over a
 * compute-dtype copy does not -- hipt_vit256_forward_range_px gives the bits of hipt_vit256_forward / hipt_hipt4k_forward on the same pixels,
 * hipt_vit256_forward_range agrees with them to the bf16 bar only.  Split a call with the _px form (uint8 input: hipt_u8_normalize to fp32 first). */
size_t hipt_vit256_range_px_workspace_bytes(const hipt_vit_weights* w, const hipt_image_layout* lay, int nseq, int chunk);
int hipt_vit256_forward_range_px(const hipt_vit_weights* w, const float* images, const hipt_image_layout* lay, int seq0, int nseq,
                                 int chunk, float* out, void* workspace, size_t ws_bytes, void* stream);

/* Whole ViT-4K forward -> out[nseq, D] fp32. */
int hipt_vit4k_forward(const hipt_vit_weights* w, const float* tokens_in, int nseq, float* out,
                       void* workspace, size_t ws_bytes, void* stream);

/* HIPT_4K.forward (HIPT_4K/hipt_4k.py:48-76) for `nreg` regions already cropped to multiples of 256:
 * regions fp32 [nreg,3,W,H] -> cls256[nreg*w_256*h_256, 384] (kept on device, no CPU hop) -> out[nreg,192].
 * The reference processes one region per call (batch_size 1, hipt_4k.py:73); nreg > 1 is the throughput
 * form: regions are independent, so their patches are simply stacked along the sequence axis.
 * cls256_out may be NULL.  workspace >= hipt_hipt4k_workspace_bytes(). */
size_t hipt_hipt4k_workspace_bytes(const hipt_vit_weights* w256, const hipt_vit_weights* w4k,
                                   int nreg, int w_256, int h_256, int chunk);
int hipt_hipt4k_forward(const hipt_vit_weights* w256, const hipt_vit_weights* w4k,
                        const float* regions, int nreg, int W, int H, int chunk, float* cls256_out, float* out,
                        void* workspace, size_t ws_bytes, void* stream);

/* uint8 input (replaces eval_transforms = ToTensor + Normalize(0.5, 0.5),
 * HIPT_4K/hipt_model_utils.py:113-118, and the float H2D copy of extract_features_fp.py:166): regions are
 * uint8 [nreg, 3, W, H] (interleaved = 0) or [nreg, W, H, 3] (interleaved = 1, the layout of a decoded RGB
 * tile); they are normalised on device, (x / 255 - 0.5) / 0.5 in fp32 exactly as torchvision, into the compute
 * dtype.  Same outputs as hipt_hipt4k_forward on the normalised float tensor, bit for bit.
 * workspace >= hipt_hipt4k_u8_workspace_bytes(). */
size_t hipt_hipt4k_u8_workspace_bytes(const hipt_vit_weights* w256, const hipt_vit_weights* w4k,
                                      int nreg, int w_256, int h_256, int chunk);
int hipt_hipt4k_forward_u8(const hipt_vit_weights* w256, const hipt_vit_weights* w4k,
                           const uint8_t* regions, int interleaved, int nreg, int W, int H, int chunk,
                           float* cls256_out, float* out, void* workspace, size_t ws_bytes, void* stream);

/* The normalisation alone: src uint8 [n_images, 3, plane] or [n_images, plane, 3] -> dst [n_images, 3, plane]
 * in dst_dtype (HIPT_F32 / HIPT_BF16); plane = W*H must be a multiple of 16. */
int hipt_u8_normalize(const void* src, int interleaved, int64_t n_images, int64_t plane, void* dst,
                      int dst_dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * CLAM_SB / ABMIL gated-attention pooling (models/model_clam.py:41-64, 77-191)
 * ---------------------------------------------------------------------------------- */
typedef struct hipt_clam_weights {
    int32_t dtype;            /* type of w1 / wab (and of the bag)                              */
    int32_t s0, s1, s2;       /* size_dict entry [S0,S1,S2] (model_clam.py:81)                  */
    int32_t n_classes;        /* bag classifier outputs                                         */
    int32_t n_att;            /* attention branches K: 0 / 1 = one (CLAM_SB; the field was `reserved` up to ABI 4); K > 1 = CLAM_MB
                                 (hipt_clam_mb_forward): wc = [K,S2], bc = [K], wcls = the K Linear(S1,1) stacked [K,S1], bcls = [K],
                                 n_classes = K, logit_bound = the largest of the K bounds */
    const void*  w1;  const float* b1;     /* attention_net.0: Linear(S0,S1) (+ReLU)  [S1,S0]     */
    const void*  wab; const float* bab;    /* attention_a.0 / attention_b.0 stacked [2*S2,S1]:
                                              rows [0,S2) = a, rows [S2,2*S2) = b; bias likewise */
    const float* wc;  const float* bc;     /* attention_c Linear(S2,1): [S2], [1]               */
    const float* wcls; const float* bcls;  /* classifiers Linear(S1,C): [C,S1], [C]             */
    float        logit_bound; /* optional (0 = unknown): an upper bound of |A_raw - bc| = sum_j |wc_j| (tanh * sigmoid lies in
                                 (-1, 1)), computed in fp32 by the owner of the weights; lets the streaming kernels exponentiate against
                                 the fixed shift bc + bound instead of a running maximum when the bound is small enough */
    int32_t      reserved2;
    const void*  stream_pk;   /* optional (NULL = absent): the weights as the LDS image of the streaming kernel (MFMA operand fragments
                                 of W1 and [Wa;Wb], biases and wc in accumulator order), written by hipt_clam_stream_pack into
                                 hipt_clam_stream_packed_bytes(w) bytes of device memory; bf16 [384|192,128,64] only.  With it (and a
                                 logit_bound < 60) hipt_clam_sb_forward is ONE launch of the streaming kernel; without it the
                                 general kernels run */
} hipt_clam_weights;

/* The streaming kernel's weight image (hipt_clam_weights.stream_pk): size in bytes (0: this shape / type has none) and the
 * packing launch (reads w1, b1, wab, bab, wc; `out` 256-byte aligned device memory).  Pack once per set of weights. */
size_t hipt_clam_stream_packed_bytes(const hipt_clam_weights* w);
int hipt_clam_stream_pack(const hipt_clam_weights* w, void* out, void* stream);

/* Scratch of the calls below.  ONE piece of state lives in it: the 256-byte "ticket block" at byte offset
 * hipt_clam_ticket_offset() (= 0: the head of the workspace, whatever the widths; the arrival counter of the in-kernel
 * combine).  It must be ZERO before the first call that uses a given workspace (zero it once when allocating) and every
 * completed call leaves it zero again -- no memset per call, any dispatch order, graph-replayable; no path of the library
 * writes anything else there, so one workspace may serve modules of different widths.  Calls that may run concurrently
 * (different streams) need different workspaces. */
size_t hipt_clam_workspace_bytes(const hipt_clam_weights* w, int N);
size_t hipt_clam_ticket_offset(const hipt_clam_weights* w, int N);

/* CLAM_SB.forward (model_clam.py:147-191, eval path without instance_eval), one bag:
 *   h1 = ReLU(bag W1^T + b1); A_raw = (tanh(h1 Wa^T+ba) * sigmoid(h1 Wb^T+bb)) wc + bc;
 *   M = softmax_N(A_raw) h1; logits = M Wcls^T + bcls; Y_prob = softmax(logits);
 *   Y_hat = argmax(logits) (int64, first maximum, as torch.topk(logits,1)).
 * bag: [N,S0] in w->dtype.  Outputs fp32: A_raw[N], M[S1], logits[C], Y_prob[C]; Y_hat int64[1].
 * attention_only != 0 skips pooling (only A_raw is written; model_clam.py:151-152). */
int hipt_clam_sb_forward(const hipt_clam_weights* w, const void* bag, int N, int attention_only,
                         float* A_raw, float* M, float* logits, float* Y_prob, int64_t* Y_hat,
                         void* workspace, size_t ws_bytes, void* stream);

/* CLAM_MB.forward (model_clam.py:226-264), inference, with ONE pass over the bag for all K = w->n_att attention branches (2 <= K <= 4; W1 and
 * [Wa; Wb] are shared by the branches, only wc / bc / the classifier rows differ):  A_raw[K, N] = the K logits of every row;
 * M[K, S1] = softmax_N(A_raw[k]) h1;  logits[K]: logits[k] = wcls[k] . M[k] + bcls[k] (:248-250; Y_prob / Y_hat are K numbers: the caller's).
 * Two launches: the streaming kernel (MFMA projections, gate once per row, K logits, h1 left in HBM as bf16) and a pooling kernel.
 * hipt_clam_mb_supported(w) != 0: this configuration has the one-pass form (bf16 [384 | 192, 128, 64], stream_pk packed with n_att = K,
 * logit_bound < 60); otherwise call hipt_clam_sb_forward once per branch.  attention_only != 0: A_raw only.
 * workspace >= hipt_clam_mb_workspace_bytes(w, N), 256-byte aligned, its first 256 bytes zero before the first use (the ticket block). */
int hipt_clam_mb_supported(const hipt_clam_weights* w);
size_t hipt_clam_mb_workspace_bytes(const hipt_clam_weights* w, int N);
int hipt_clam_mb_forward(const hipt_clam_weights* w, const void* bag, int N, int attention_only, float* A_raw, float* M, float* logits,
                         void* workspace, size_t ws_bytes, void* stream);

/* Attn_Net_Gated.forward (model_clam.py:59-64) on its own: A[N] from x[N,L]
 * (w->s1 = L, w->s2 = D; w1/b1/wcls unused). */
int hipt_attn_net_gated(const hipt_clam_weights* w, const void* x, int N, float* A,
                        void* workspace, size_t ws_bytes, void* stream);

/* inst_eval support (model_clam.py:116-145): h1 rows for `n_idx` selected instances,
 * out[n_idx, S1] fp32 = ReLU(bag[idx] W1^T + b1). */
int hipt_clam_gather_h1(const hipt_clam_weights* w, const void* bag, const int64_t* idx, int n_idx,
                        float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * CLAM training step: differentiable forward + backward of CLAM_SB.forward / CLAM_MB.forward
 * (models/model_clam.py:147-191, 226-264) with the instance branch's top-k (inst_eval / inst_eval_out, :116-145), as
 * driven by utils/core_utils.py:300-348 (train_loop_clam) and :373-426 (train_loop; loss.backward() at :423).
 * fp32 throughout (the reference's training precision).  A forward is two launches, a backward two.
 * ---------------------------------------------------------------------------------- */
typedef struct hipt_clam_train_weights {
    int32_t s0, s1, s2;       /* size_dict entry; each a multiple of 4 (every entry of the reference's table is)  */
    int32_t n_att;            /* K attention branches: 1 (CLAM_SB) or n_classes (CLAM_MB); <= 8                   */
    int32_t n_classes;        /* C <= 8                                                                           */
    int32_t multi_branch;     /* 0: logits = wcls[C,S1] M[0] + bcls (CLAM_SB :181); 1: logits[c] = wcls[c] . M[c] + bcls[c] (CLAM_MB :248-250) */
    const float* w1;  const float* b1;     /* attention_net.0                     [S1,S0], [S1] */
    const float* wa;  const float* ba;     /* attention_a.0                       [S2,S1], [S2] */
    const float* wb;  const float* bb;     /* attention_b.0                       [S2,S1], [S2] */
    const float* wc;  const float* bc;     /* attention_c                         [K,S2],  [K]  */
    const float* wcls; const float* bcls;  /* classifiers (CLAM_MB: the C Linear(S1,1) stacked) [C,S1], [C] */
} hipt_clam_train_weights;

typedef struct hipt_clam_train_grads {     /* outputs of the backward, same shapes as the weights; dbag may be NULL */
    float* dw1; float* db1; float* dwa; float* dba; float* dwb; float* dbb; float* dwc; float* dbc;
    float* dwcls; float* dbcls;
    float* dbag;                           /* [N,S0] or NULL (bags normally do not require a gradient)            */
} hipt_clam_train_grads;

size_t hipt_clam_train_workspace_bytes(const hipt_clam_train_weights* w, int N);   /* scratch of the backward */
/* 1 when the training kernels take this size_dict entry with K = n_att attention branches and C = n_classes (widths multiples
 * of 4, K and C <= 8, the row tiles of forward AND backward within the 160 KiB of LDS; need_dbag: the bag requires a
 * gradient too), else 0: the caller then keeps the PyTorch-op sequence ('big' [1024,512,384] with a bag gradient does not fit). */
int hipt_clam_train_shape_supported(int s0, int s1, int s2, int n_att, int n_classes, int need_dbag);

/* Forward.  bag [N,S0] fp32.  m1 [N,S1], ma / mb [N,S2]: dropout masks already scaled (0 or 1/(1-p)), or NULL
 * (nn.Dropout after the ReLU, :86-87, and inside Attn_Net_Gated, :48-52; the caller draws them).
 * Saved for the backward (caller-allocated): h1 [N,S1] (after ReLU and dropout = the `h` the reference returns),
 * t = tanh(.) and s = sigmoid(.) [N,S2] before dropout, A_raw [K,N], stats [K,2] = (max, sum exp) of the softmax over N,
 * M [K,S1].  Results: logits [C], Y_prob [C], Y_hat int64 [1].
 * k_sample > 0: topk_ids int64 [K,2,k_sample] = per branch the ids of the k largest and of the k smallest attention
 * scores (torch.topk(A, k) / torch.topk(-A, k), :120-123; ties: lowest index first), and, when h1_sel != NULL,
 * h1_sel [K,2,k_sample,S1] = those rows of h1 (index_select).  k_sample > N is an error, as in torch.topk. */
int hipt_clam_train_forward(const hipt_clam_train_weights* w, const float* bag, int N,
                            const float* m1, const float* ma, const float* mb,
                            float* h1, float* t, float* s, float* A_raw, float* stats, float* M,
                            float* logits, float* Y_prob, int64_t* Y_hat,
                            int k_sample, int64_t* topk_ids, float* h1_sel, void* stream);

/* Backward.  Incoming gradients: dlogits [C]; optional (NULL = zero) dA_raw [K,N] and dM [K,S1] (the 'features' output);
 * optional instance-branch rows: n_sel row ids sel_ids [n_sel] with their gradients dh1_sel [n_sel,S1], added to dh1.
 * Everything else is what the forward saved.  Writes every member of g.  workspace >= hipt_clam_train_workspace_bytes(). */
int hipt_clam_train_backward(const hipt_clam_train_weights* w, const float* bag, int N,
                             const float* m1, const float* ma, const float* mb,
                             const float* h1, const float* t, const float* s, const float* A_raw,
                             const float* stats, const float* M,
                             const float* dlogits, const float* dA_raw, const float* dM,
                             const int64_t* sel_ids, const float* dh1_sel, int n_sel,
                             const hipt_clam_train_grads* g, void* workspace, size_t ws_bytes, void* stream);

/* torch.topk(A, k) and torch.topk(-A, k) of every row of A [rows, N] on the device (inst_eval, :120-123):
 * ids int64 [rows, 2, k], descending / ascending by value, ties -> lowest index first. */
int hipt_topk_rows(const float* A, int rows, int N, int k, int64_t* ids, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIPT_ABMIL_H */
