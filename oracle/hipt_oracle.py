"""CPU ORACLE (test infrastructure, NOT product code).

A plain-numpy restatement of the reference's algorithm for the one hot path this
repository accelerates (HIPT_4K feature extraction + CLAM_SB gated-attention pooling).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker; the product package never does.

Every function cites the reference file:line it follows (paths relative to the
reference checkout, scjjb/HIPT_ABMIL_ATEC23).  The arithmetic itself lives in PyTorch
(pinned ``torch==1.13.1``, requirements.txt:17): Linear = ``x @ W.T + b``, LayerNorm with
biased variance, exact-erf GELU, max-subtracted softmax, bicubic ``F.interpolate``
(align_corners=False, A=-0.75) driven by ``scale_factor``.

PARITY PIN: the reference has no tests or golden vectors for this path,
so the oracle is pinned against outputs of the reference's own modules run in the build
container: ``tests/golden/make_golden.py`` imports ``HIPT_4K.vision_transformer``,
``HIPT_4K.vision_transformer4k`` and ``models.model_clam`` from the reference checkout,
loads hash-generated weights and stores their outputs under ``tests/golden/``;
``tests/test_oracle_vs_golden.py`` checks every function here against those files.

All functions take and return numpy arrays.  ``params`` is a dict keyed by the
reference's state-dict names.  Computation happens in the dtype of the input
(fp32 to mirror the reference, fp64 when a higher-precision truth is wanted).
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf

LN_EPS = 1e-6  # partial(nn.LayerNorm, eps=1e-6): vision_transformer.py:282-286, vision_transformer4k.py:267-272


# --------------------------------------------------------------------------------------
# Elementary ops (torch.nn semantics)
# --------------------------------------------------------------------------------------

def linear(x, w, b=None):
    """nn.Linear: ``x @ w.T + b`` (w is [out, in])."""
    y = x @ w.T.astype(x.dtype)
    if b is not None:
        y = y + b.astype(x.dtype)
    return y


def layer_norm(x, w, b, eps=LN_EPS):
    """nn.LayerNorm over the last dim, biased variance (vision_transformer.py:138,142,195)."""
    mu = x.mean(axis=-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(axis=-1, keepdims=True)
    return xc / np.sqrt(var + x.dtype.type(eps)) * w.astype(x.dtype) + b.astype(x.dtype)


def gelu(x):
    """nn.GELU() default = exact erf form (vision_transformer.py:89,94)."""
    return (x * x.dtype.type(0.5) * (x.dtype.type(1.0) + _erf(x / x.dtype.type(math.sqrt(2.0))))).astype(x.dtype)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


# --------------------------------------------------------------------------------------
# Transformer block (vision_transformer.py:88-152, duplicated at vision_transformer4k.py:94-158)
# --------------------------------------------------------------------------------------

def attention(x, p, pre, num_heads):
    """Attention.forward (vision_transformer.py:119-131). Returns (y [B,N,C], attn [B,H,N,N])."""
    B, N, C = x.shape
    dh = C // num_heads
    scale = x.dtype.type(dh ** -0.5)  # :112
    qkv = linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"])  # :121
    qkv = qkv.reshape(B, N, 3, num_heads, dh).transpose(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]  # each [B,H,N,dh]
    attn = (q @ k.transpose(0, 1, 3, 2)) * scale  # :124
    attn = softmax(attn, axis=-1)  # :125
    y = (attn @ v).transpose(0, 2, 1, 3).reshape(B, N, C)  # :128
    y = linear(y, p[pre + "proj.weight"], p[pre + "proj.bias"])  # :129
    return y, attn


def mlp(x, p, pre):
    """Mlp.forward (vision_transformer.py:98-104); dropouts are identity (p=0, eval)."""
    h = gelu(linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"]))
    return linear(h, p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def block(x, p, i, num_heads, return_attention=False):
    """Block.forward (vision_transformer.py:146-152)."""
    pre = f"blocks.{i}."
    y, attn = attention(layer_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]), p, pre + "attn.", num_heads)
    if return_attention:
        return attn
    x = x + y
    x = x + mlp(layer_norm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp.")
    return x


def _depth(p):
    return 1 + max(int(k.split(".")[1]) for k in p if k.startswith("blocks."))


# --------------------------------------------------------------------------------------
# Positional-embedding interpolation (vision_transformer.py:213-233, vision_transformer4k.py:201-221)
# --------------------------------------------------------------------------------------

def _cubic_coeffs(t, A=-0.75):
    """ATen get_cubic_upsample_coefficients (UpSample.h), A = -0.75."""
    def c1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def c2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], axis=-1)


def _bicubic_axis(n_in, n_out, scale_factor):
    """Per-output-index source taps and weights for ``F.interpolate(mode='bicubic',
    align_corners=False, scale_factor=s)``: the GIVEN scale factor (not out/in) maps
    destination to source coordinates, ``src = (dst + 0.5) / s - 0.5`` (no clamping for
    cubic), taps ``floor(src) - 1 .. + 2`` clamped to the border."""
    inv = np.float32(1.0 / scale_factor)
    dst = np.arange(n_out, dtype=np.float32)
    src = inv * (dst + np.float32(0.5)) - np.float32(0.5)
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    t = np.clip(src - i0.astype(np.float32), 0.0, 1.0).astype(np.float32)
    w = _cubic_coeffs(t.astype(np.float64))  # [n_out, 4]
    idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)  # [n_out, 4]
    return idx, w


def interpolate_pos_encoding(pos_embed, npatch, w, h, patch_size):
    """``interpolate_pos_encoding(x, w, h)``; ``w``/``h`` are dims 2/3 of the model input.

    pos_embed: [1, 1+N, D] -> [1, 1+w0*h0, D] with w0 = w // patch_size, h0 = h // patch_size.
    The identity shortcut (``npatch == N and w == h``) is kept (vision_transformer.py:216-217).
    """
    N = pos_embed.shape[1] - 1
    if npatch == N and w == h:
        return pos_embed
    D = pos_embed.shape[-1]
    g = int(math.sqrt(N))
    w0, h0 = w // patch_size, h // patch_size
    sw, sh = (w0 + 0.1) / math.sqrt(N), (h0 + 0.1) / math.sqrt(N)  # :224,228
    ow, oh = int(math.floor(g * sw)), int(math.floor(g * sh))
    assert ow == w0 and oh == h0  # :231
    grid = pos_embed[0, 1:].reshape(g, g, D).astype(np.float64)  # [i (dim2), j (dim3), D]
    ii, wi = _bicubic_axis(g, ow, sw)
    jj, wj = _bicubic_axis(g, oh, sh)
    # separable: rows (dim 2) then columns (dim 3)
    tmp = np.einsum("oa,oajd->ojd", wi, grid[ii])  # [ow, g, D]
    out = np.einsum("pb,opbd->opd", wj, tmp[:, jj])  # [ow, oh, D]
    out = out.reshape(1, ow * oh, D).astype(pos_embed.dtype)
    return np.concatenate([pos_embed[:, :1], out], axis=1)


# --------------------------------------------------------------------------------------
# ViT-256 (vision_transformer.py:155-272)
# --------------------------------------------------------------------------------------

def patch_embed(x, w, b, patch_size=16):
    """PatchEmbed.forward (vision_transformer.py:167-170): Conv2d(k=s=patch) as a GEMM.
    x [B,C,W,H] -> [B, (W/ps)*(H/ps), D]; token = i*(H/ps)+j, k = c*ps*ps + ky*ps + kx."""
    B, C, W, H = x.shape
    ps = patch_size
    nw, nh = W // ps, H // ps
    xp = x[:, :, : nw * ps, : nh * ps].reshape(B, C, nw, ps, nh, ps).transpose(0, 2, 4, 1, 3, 5)
    xp = xp.reshape(B, nw * nh, C * ps * ps)
    return linear(xp, w.reshape(w.shape[0], -1), b)


def vit256_prepare_tokens(x, p, patch_size=16):
    """VisionTransformer.prepare_tokens (vision_transformer.py:235-246)."""
    B, _, w, h = x.shape
    t = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], patch_size)
    cls = np.broadcast_to(p["cls_token"].astype(x.dtype), (B, 1, t.shape[-1]))
    t = np.concatenate([cls, t], axis=1)
    pos = interpolate_pos_encoding(p["pos_embed"], t.shape[1] - 1, w, h, patch_size).astype(x.dtype)
    return t + pos


def _run_blocks(x, p, num_heads, taps=None):
    out = {}
    for i in range(_depth(p)):
        x = block(x, p, i, num_heads)
        if taps and i in taps:
            out[i] = x
    return x, out


def vit256_forward(x, p, num_heads=6, patch_size=16):
    """VisionTransformer.forward (vision_transformer.py:248-253): CLS row of the final LN."""
    t = vit256_prepare_tokens(x, p, patch_size)
    t, _ = _run_blocks(t, p, num_heads)
    t = layer_norm(t, p["norm.weight"], p["norm.bias"])
    return t[:, 0]


def vit_last_selfattention(tokens, p, num_heads):
    """get_last_selfattention after prepare_tokens (vision_transformer.py:255-262)."""
    d = _depth(p)
    x = tokens
    for i in range(d - 1):
        x = block(x, p, i, num_heads)
    return block(x, p, d - 1, num_heads, return_attention=True)


def vit_intermediate_layers(tokens, p, num_heads, n=1):
    """get_intermediate_layers after prepare_tokens (vision_transformer.py:264-272)."""
    d = _depth(p)
    x, outs = tokens, []
    for i in range(d):
        x = block(x, p, i, num_heads)
        if d - i <= n:
            outs.append(layer_norm(x, p["norm.weight"], p["norm.bias"]))
    return outs


# --------------------------------------------------------------------------------------
# ViT-4K (vision_transformer4k.py:161-265)
# --------------------------------------------------------------------------------------

def vit4k_prepare_tokens(x, p):
    """VisionTransformer4K.prepare_tokens (vision_transformer4k.py:223-239).
    x [B, 384, w, h] -> tokens [B, 1+w*h, 192]; patch size for pos interpolation is 1 (:208-209)."""
    B, E, w, h = x.shape
    t = x.reshape(B, E, w * h).transpose(0, 2, 1)
    t = gelu(linear(t, p["phi.0.weight"], p["phi.0.bias"]))  # phi = Linear + GELU (+Dropout p=0), :169
    cls = np.broadcast_to(p["cls_token"].astype(x.dtype), (B, 1, t.shape[-1]))
    t = np.concatenate([cls, t], axis=1)
    pos = interpolate_pos_encoding(p["pos_embed"], t.shape[1] - 1, w, h, 1).astype(x.dtype)
    return t + pos


def vit4k_forward(x, p, num_heads=6):
    """VisionTransformer4K.forward (vision_transformer4k.py:241-246)."""
    t = vit4k_prepare_tokens(x, p)
    t, _ = _run_blocks(t, p, num_heads)
    t = layer_norm(t, p["norm.weight"], p["norm.bias"])
    return t[:, 0]


# --------------------------------------------------------------------------------------
# HIPT_4K wrapper (hipt_4k.py:48-76, 308-330)
# --------------------------------------------------------------------------------------

def center_crop_offsets(size, crop):
    """torchvision CenterCrop offset: ``int(round((size - crop) / 2.0))`` (Python round =
    half-to-even); used by prepare_img_tensor (hipt_4k.py:308-330).  torchvision is absent in
    the build container, so this one line is restated from torchvision 0.13 and is NOT pinned
    by a golden; it is the identity for sizes divisible by 256 (the BASELINE configs)."""
    return int(round((size - crop) / 2.0))


def prepare_img_tensor(img, patch_size=256):
    """hipt_4k.py:308-330 -> (cropped img, w_256, h_256)."""
    _, _, w, h = img.shape
    W, H = w - w % patch_size, h - h % patch_size
    t, l = center_crop_offsets(w, W), center_crop_offsets(h, H)
    return img[:, :, t:t + W, l:l + H], w // patch_size, h // patch_size


def patchify_256(img, w_256, h_256):
    """unfold(2,256,256).unfold(3,256,256) + 'b c p1 p2 w h -> (b p1 p2) c w h'
    (hipt_4k.py:64-65): patch k = p1*h_256 + p2, row-major."""
    b, c, _, _ = img.shape
    x = img.reshape(b, c, w_256, 256, h_256, 256).transpose(0, 2, 4, 1, 3, 5)
    return x.reshape(b * w_256 * h_256, c, 256, 256)


def cls_grid(features_cls256, w_256, h_256):
    """reshape(w,h,384).transpose(0,1).transpose(0,2).unsqueeze(0) (hipt_4k.py:73):
    [B,384] -> [1,384,w_256,h_256] with grid[0,:,i,j] = features[i*h_256 + j]."""
    g = features_cls256.reshape(w_256, h_256, -1)
    g = g.transpose(1, 0, 2).transpose(2, 1, 0)
    return g[None]


def hipt4k_forward(x, p256, p4k, heads256=6, heads4k=6, return_cls256=False):
    """HIPT_4K.forward (hipt_4k.py:48-76); minibatches of 256 patches (:68)."""
    img, w_256, h_256 = prepare_img_tensor(x)
    batch = patchify_256(img, w_256, h_256)
    feats = [vit256_forward(batch[i:i + 256], p256, heads256) for i in range(0, batch.shape[0], 256)]
    f256 = np.concatenate(feats, axis=0)
    out = vit4k_forward(cls_grid(f256, w_256, h_256), p4k, heads4k)
    return (out, f256) if return_cls256 else out


def nearest_upsample(a, f):
    """F.interpolate(a, scale_factor=f, mode="nearest") for an integer factor on the two last axes: out[i, j] = a[i // f, j // f]."""
    return np.repeat(np.repeat(a, f, axis=-2), f, axis=-1)


def region_attention_scores(x, p256, p4k, scale=1, heads256=6, heads4k=6):
    """The tensor half of HIPT_4K._get_region_attention_scores (hipt_4k.py:135-160) for a normalised region x [1,3,W,H]:
    returns (patches [n,3,256/s,256/s] float, attention_256 [n,heads,256/s,256/s], attention_4k [heads,W/s,H/s]).
    attention_* are the [CLS] query's row of the last block's attention map without its own column ([:, :, 0, 1:]), laid out on
    the token grid and blown up with nearest-neighbour copies by int(16/scale) / int(256/scale)."""
    img, w_256, h_256 = prepare_img_tensor(x)
    batch = patchify_256(img, w_256, h_256)  # :138-139
    n = batch.shape[0]
    tok = vit256_prepare_tokens(batch, p256)
    a256 = vit_last_selfattention(tok, p256, heads256)[:, :, 0, 1:]  # :143-145
    a256 = nearest_upsample(a256.reshape(n, heads256, 16, 16), int(16 / scale))  # :146-147
    f256 = vit256_forward(batch, p256, heads256)  # :141
    grid = cls_grid(f256, w_256, h_256)  # :149
    a4k = vit_last_selfattention(vit4k_prepare_tokens(grid, p4k), p4k, heads4k)[0, :, 0, 1:]  # :153-155
    a4k = nearest_upsample(a4k.reshape(heads4k, w_256, h_256), int(256 / scale))  # :156-157
    if scale != 1:  # :159-160: F.interpolate(scale_factor=1/scale, nearest) on the patches: out[i] = in[floor(i * scale)]
        idx = np.floor(np.arange(int(256 * (1 / scale))) * scale).astype(np.int64)
        batch = batch[:, :, idx][:, :, :, idx]
    return batch, a256, a4k


# --------------------------------------------------------------------------------------
# CLAM / ABMIL aggregator (models/model_clam.py)
# --------------------------------------------------------------------------------------

def _gate_index(p):
    for k in p:
        if k.startswith("attention_net.") and ".attention_a." in k:
            return int(k.split(".")[1])
    raise KeyError("no gated attention in params")


def attn_net_gated(x, p, pre=""):
    """Attn_Net_Gated.forward (model_clam.py:59-64) -> (A [N,K], x)."""
    a = np.tanh(linear(x, p[pre + "attention_a.0.weight"], p[pre + "attention_a.0.bias"]))
    z = linear(x, p[pre + "attention_b.0.weight"], p[pre + "attention_b.0.bias"])
    b = x.dtype.type(1.0) / (x.dtype.type(1.0) + np.exp(-z))
    A = linear(a * b, p[pre + "attention_c.weight"], p[pre + "attention_c.bias"])
    return A, x


def topk_desc(v, k):
    """torch.topk(v, k)[1]: indices of the k largest, descending (ties: lowest index first)."""
    return np.argsort(-v, kind="stable")[:k]


def clam_sb_forward(h, p, k_sample=8, label=None, instance_eval=False, subtyping=False,
                    attention_only=False):
    """CLAM_SB.forward (model_clam.py:147-191), eval mode (dropouts identity).

    Returns dict(logits [1,C], Y_prob [1,C], Y_hat [1,1] int64, A_raw [1,N], M [1,S1]) plus, with
    instance_eval, ``inst_ids`` (top-k ids per evaluated branch), ``inst_logits`` and
    ``instance_loss`` (CE, mean reduction) following inst_eval/inst_eval_out (:116-145)."""
    g = _gate_index(p)
    h1 = np.maximum(linear(h, p["attention_net.0.weight"], p["attention_net.0.bias"]), 0)  # :83
    A, _ = attn_net_gated(h1, p, f"attention_net.{g}.")
    A = A.T  # :150
    if attention_only:
        return A
    A_raw = A
    A = softmax(A, axis=1)  # :154
    res = {}
    if instance_eval:
        n_classes = p["classifiers.weight"].shape[0]
        total, ids, ilogits = 0.0, [], []
        for c in range(n_classes):
            w, b = p[f"instance_classifiers.{c}.weight"], p[f"instance_classifiers.{c}.bias"]
            if int(label) == c:  # in-the-class (:116-132)
                tp, tn = topk_desc(A[-1], k_sample), topk_desc(-A[-1], k_sample)
                inst = np.concatenate([h1[tp], h1[tn]], axis=0)
                tgt = np.concatenate([np.ones(k_sample, np.int64), np.zeros(k_sample, np.int64)])
                ids.append(np.concatenate([tp, tn]))
            elif subtyping:  # out-of-the-class (:135-145)
                tp = topk_desc(A[-1], k_sample)
                inst, tgt = h1[tp], np.zeros(k_sample, np.int64)
                ids.append(tp)
            else:
                continue
            lg = linear(inst, w, b)
            ls = lg - lg.max(axis=1, keepdims=True)
            lse = np.log(np.exp(ls).sum(axis=1))
            total += float(np.mean(lse - ls[np.arange(len(tgt)), tgt]))
            ilogits.append(lg)
        if subtyping:
            total /= n_classes
        res.update(instance_loss=total, inst_ids=ids, inst_logits=ilogits)
    M = A @ h1  # :180
    logits = linear(M, p["classifiers.weight"], p["classifiers.bias"])  # :181
    Y_hat = topk_desc(logits[0], 1).reshape(1, 1).astype(np.int64)  # :182
    Y_prob = softmax(logits, axis=1)  # :183
    res.update(logits=logits, Y_prob=Y_prob, Y_hat=Y_hat, A_raw=A_raw, M=M)
    return res
