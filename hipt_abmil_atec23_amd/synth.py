"""Deterministic, bit-reproducible synthetic data for tests, goldens and benchmarks.

There is no network (no pretrained DINO checkpoints, no slides), so every weight and
input on the hot path comes from one integer hash: element ``i`` of the tensor with
seed ``s`` is ``fmix32(i * 0x9E3779B1 + s * 0x85EBCA77 + 0x165667B1)``, mapped to
``[-1, 1)`` with a 2**-23 step (exact in fp32) and then scaled/shifted in fp32.  The
same bits come out of numpy on the CPU and of torch on any device, so golden outputs
can be committed without committing their inputs.

The value ranges follow the reference initialisers: ViT weights are
``trunc_normal_(std=.02)`` (HIPT_4K/vision_transformer.py:204-211) -> uniform with a
comparable spread; CLAM weights are xavier-normal with zero bias
(utils/utils.py:217-225) -> uniform with the same standard deviation.  Biases and LN
affine terms are made non-trivial on purpose so that every bias path is exercised.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np

_MASK = 0xFFFFFFFF
_C_IDX, _C_SEED, _C_ADD = 0x9E3779B1, 0x85EBCA77, 0x165667B1
_M1, _M2 = 0x85EBCA6B, 0xC2B2AE35


def hash_u32_np(n: int, seed: int, start: int = 0) -> np.ndarray:
    """uint32 hash of indices ``start .. start+n-1`` (numpy, CPU)."""
    h = np.arange(start, start + n, dtype=np.uint64)
    h = (h * np.uint64(_C_IDX) + np.uint64((seed * _C_SEED + _C_ADD) & _MASK)) & np.uint64(_MASK)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(_M1)) & np.uint64(_MASK)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(_M2)) & np.uint64(_MASK)
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def hash_uniform_np(shape, seed: int, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """fp32 array, ``offset + scale * u`` with ``u`` uniform on [-1, 1) (numpy)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (hash_u32_np(n, seed) >> np.uint32(8)).astype(np.float32)
    f = u * np.float32(2.0 ** -23) - np.float32(1.0)
    f = f * np.float32(scale)
    if offset != 0.0:
        f = f + np.float32(offset)
    return f.reshape(shape)


def hash_uniform_torch(shape, seed: int, scale: float = 1.0, offset: float = 0.0, device="cpu"):
    """Same bits as :func:`hash_uniform_np`, generated with torch ops on ``device``."""
    import torch

    n = 1
    for s in shape:
        n *= int(s)
    h = torch.arange(n, dtype=torch.int64, device=device)
    h = (h * _C_IDX + ((seed * _C_SEED + _C_ADD) & _MASK)) & _MASK
    h = h ^ (h >> 16)
    h = (h * _M1) & _MASK
    h = h ^ (h >> 13)
    h = (h * _M2) & _MASK
    h = h ^ (h >> 16)
    f = (h >> 8).to(torch.float32) * (2.0 ** -23) - 1.0
    f = f * scale
    if offset != 0.0:
        f = f + offset
    return f.reshape(tuple(shape))


def name_seed(name: str, base: int = 0) -> int:
    """Stable per-tensor seed from the parameter name."""
    return (zlib.crc32(name.encode()) + 7919 * base) & 0x7FFFFFFF


# ----------------------------------------------------------------------------------
# Parameter specs.  Shapes mirror the reference state-dict layout.
# ----------------------------------------------------------------------------------

def vit_param_specs(kind: str = "vit256", embed_dim: int = 384, depth: int = 12, num_heads: int = 6,
                    in_dim: int = 384, patch_size: int = 16, in_chans: int = 3, n_pos: int = 197,
                    mlp_ratio: float = 4.0):
    """``name -> (shape, scale, offset)`` for a ViT-256 (``kind='vit256'``) or ViT-4K."""
    D = embed_dim
    hid = int(D * mlp_ratio)
    sp = OrderedDict()
    sp["cls_token"] = ((1, 1, D), 0.035, 0.0)
    sp["pos_embed"] = ((1, n_pos, D), 0.035, 0.0)
    if kind == "vit256":
        sp["patch_embed.proj.weight"] = ((D, in_chans, patch_size, patch_size), 0.035, 0.0)
        sp["patch_embed.proj.bias"] = ((D,), 0.02, 0.0)
    else:
        sp["phi.0.weight"] = ((D, in_dim), 0.035, 0.0)
        sp["phi.0.bias"] = ((D,), 0.02, 0.0)
    for i in range(depth):
        p = f"blocks.{i}."
        sp[p + "norm1.weight"] = ((D,), 0.1, 1.0)
        sp[p + "norm1.bias"] = ((D,), 0.05, 0.0)
        # qkv weights are ~3x the trunc_normal spread so that softmax rows are far
        # from uniform (a uniform row would hide masking / max-subtraction bugs)
        sp[p + "attn.qkv.weight"] = ((3 * D, D), 0.1, 0.0)
        sp[p + "attn.qkv.bias"] = ((3 * D,), 0.02, 0.0)
        sp[p + "attn.proj.weight"] = ((D, D), 0.035, 0.0)
        sp[p + "attn.proj.bias"] = ((D,), 0.02, 0.0)
        sp[p + "norm2.weight"] = ((D,), 0.1, 1.0)
        sp[p + "norm2.bias"] = ((D,), 0.05, 0.0)
        sp[p + "mlp.fc1.weight"] = ((hid, D), 0.035, 0.0)
        sp[p + "mlp.fc1.bias"] = ((hid,), 0.02, 0.0)
        sp[p + "mlp.fc2.weight"] = ((D, hid), 0.035, 0.0)
        sp[p + "mlp.fc2.bias"] = ((D,), 0.02, 0.0)
    sp["norm.weight"] = ((D,), 0.1, 1.0)
    sp["norm.bias"] = ((D,), 0.05, 0.0)
    return sp


def clam_param_specs(size=(384, 128, 64), n_classes: int = 2, dropout: bool = False, k_attn: int = 1, multi: bool = False):
    """``name -> (shape, scale, offset)`` for CLAM_SB (models/model_clam.py:77-100); ``multi``: CLAM_MB (:193-224), one
    attention branch and one ``Linear(S1, 1)`` bag classifier per class."""
    s0, s1, s2 = size
    if multi:
        k_attn = n_classes
    g = 3 if dropout else 2  # index of Attn_Net_Gated inside attention_net (Dropout shifts it)

    def xav(fo, fi):
        return math.sqrt(2.0 / (fi + fo)) * math.sqrt(3.0)

    sp = OrderedDict()
    sp["attention_net.0.weight"] = ((s1, s0), xav(s1, s0), 0.0)
    sp["attention_net.0.bias"] = ((s1,), 0.02, 0.0)
    sp[f"attention_net.{g}.attention_a.0.weight"] = ((s2, s1), xav(s2, s1), 0.0)
    sp[f"attention_net.{g}.attention_a.0.bias"] = ((s2,), 0.02, 0.0)
    sp[f"attention_net.{g}.attention_b.0.weight"] = ((s2, s1), xav(s2, s1), 0.0)
    sp[f"attention_net.{g}.attention_b.0.bias"] = ((s2,), 0.02, 0.0)
    # attention_c is scaled up so that the softmax over N is peaked rather than flat
    sp[f"attention_net.{g}.attention_c.weight"] = ((k_attn, s2), 4.0 * xav(k_attn, s2), 0.0)
    sp[f"attention_net.{g}.attention_c.bias"] = ((k_attn,), 0.02, 0.0)
    if multi:
        for c in range(n_classes):
            sp[f"classifiers.{c}.weight"] = ((1, s1), xav(n_classes, s1), 0.0)
            sp[f"classifiers.{c}.bias"] = ((1,), 0.02, 0.0)
    else:
        sp["classifiers.weight"] = ((n_classes, s1), xav(n_classes, s1), 0.0)
        sp["classifiers.bias"] = ((n_classes,), 0.02, 0.0)
    for c in range(n_classes):
        sp[f"instance_classifiers.{c}.weight"] = ((2, s1), xav(2, s1), 0.0)
        sp[f"instance_classifiers.{c}.bias"] = ((2,), 0.02, 0.0)
    return sp


def make_params_np(specs, base_seed: int = 0):
    """Materialise a spec dict as fp32 numpy arrays."""
    return OrderedDict((k, hash_uniform_np(shape, name_seed(k, base_seed), sc, off))
                       for k, (shape, sc, off) in specs.items())


def make_state_dict(specs, base_seed: int = 0, device="cpu"):
    """Materialise a spec dict as torch tensors (same bits as :func:`make_params_np`)."""
    return OrderedDict((k, hash_uniform_torch(shape, name_seed(k, base_seed), sc, off, device=device))
                       for k, (shape, sc, off) in specs.items())


# ----------------------------------------------------------------------------------
# The OUTLIER weight family (round 6): the benign family above has LN gamma ~ 1 and O(1) activations everywhere; trained DINO ViTs
# do not -- a few residual channels carry magnitudes of 10^2 ("massive activations"), LayerNorm gains spread over two decades, some
# heads are peaky.  The same specs, then deterministic edits (exactly representable constants, chosen by the hash):
#   * every LayerNorm gain: ~2 % of the channels take a value from GAINS (0.05 ... 20);
#   * four residual channels get a bias of +-(50 ... 100) in the patch embedding (ViT-256: patch_embed.proj.bias; ViT-4K: phi.0.bias)
#     and four more in blocks.0.mlp.fc2.bias;
#   * in every block the q and k rows of ONE head (block i: head i mod heads) are scaled by `qk_scale` each: logits of that head ~ +-40.
# ----------------------------------------------------------------------------------
GAINS = (0.05, 0.1, 0.25, 0.5, 2.0, 4.0, 8.0, 20.0)
_BIG = (50.0, -62.5, 75.0, -100.0)


def apply_vit_outliers_np(params, heads: int, qk_scale: float = 3.0, seed: int = 77):
    """In-place edits of a ``make_params_np`` dict of a ViT (see above); returns it."""
    D = params["norm.weight"].shape[0]
    dh = D // heads
    for k in list(params):
        if k.endswith(("norm1.weight", "norm2.weight")) or k == "norm.weight":
            h = hash_u32_np(D, name_seed(k, seed))
            idx = np.nonzero(h % np.uint32(50) == 0)[0]
            params[k][idx] = np.asarray(GAINS, np.float32)[(h[idx] >> np.uint32(8)) % np.uint32(len(GAINS))]
    ek = "patch_embed.proj.bias" if "patch_embed.proj.bias" in params else "phi.0.bias"
    for j, k in enumerate((ek, "blocks.0.mlp.fc2.bias")):
        ch = np.unique(hash_u32_np(4, name_seed(k, seed + 1)) % np.uint32(D))
        params[k][ch] = np.asarray(_BIG, np.float32)[:len(ch)] * np.float32(1.0 if j == 0 else -1.0)
    i = 0
    while f"blocks.{i}.attn.qkv.weight" in params:
        w, b = params[f"blocks.{i}.attn.qkv.weight"], params[f"blocks.{i}.attn.qkv.bias"]
        hd = i % heads
        for part in (0, 1):  # q rows, k rows
            sl = slice(part * D + hd * dh, part * D + (hd + 1) * dh)
            w[sl] *= np.float32(qk_scale)
            b[sl] *= np.float32(qk_scale)
        i += 1
    return params


def make_vit_outlier_params_np(specs, base_seed: int, heads: int, qk_scale: float = 3.0):
    return apply_vit_outliers_np(make_params_np(specs, base_seed), heads, qk_scale)


def make_vit_outlier_state_dict(specs, base_seed: int, heads: int, qk_scale: float = 3.0, device="cpu"):
    """torch tensors with the bits of :func:`make_vit_outlier_params_np` (made on the host, then moved)."""
    import torch
    return OrderedDict((k, torch.from_numpy(v).to(device)) for k, v in make_vit_outlier_params_np(specs, base_seed, heads, qk_scale).items())


def scale_clam_attention_c_np(params, target_bound: float):
    """attention_c.weight rescaled so that sum_j |wc_j| (the bound of |A_raw - bc|: tanh * sigmoid is inside (-1, 1)) equals
    `target_bound` up to fp32 rounding -- CLAM's fixed-shift softmax kernel runs below 60, the general kernels above."""
    k = [n for n in params if n.endswith("attention_c.weight")][0]
    params[k] = params[k] * np.float32(target_bound / float(np.abs(params[k].astype(np.float64)).sum(axis=1).max()))
    return params
