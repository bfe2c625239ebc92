"""ViT-256 host mirror: the reference's ``HIPT_4K/vision_transformer.py`` call surface on top of
the gfx950 library.

Same class names, constructor arguments, attribute names and state-dict keys as the reference
(``cls_token``, ``pos_embed``, ``patch_embed.proj.*``, ``blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,
mlp.fc1,mlp.fc2}.*``, ``norm.*``) so DINO checkpoints load unchanged.  The modules
only HOLD parameters; every forward is executed by hand-written HIP kernels through
``libhipt_abmil.so``.  There is no CPU path: a CPU input raises.

``compute_dtype``: ``'fp32'`` (default — exact-fp32 MFMA, reference numerics) or ``'bf16'``
(bf16 MFMA operands, fp32 accumulate / LayerNorm / softmax / residual stream); set with
``model.set_compute_dtype('bf16')`` or the ``HIPT_AMD_DTYPE`` environment variable.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from functools import partial

import torch
import torch.nn as nn

from . import _native as N
from . import functional as Fn


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """Truncated-normal initialiser with the reference's argument meaning
    (vision_transformer.py:63-65)."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _default_dtype() -> str:
    return os.environ.get("HIPT_AMD_DTYPE", "fp32")


class _PackedVit:
    """Device-side weight image of one ViT for one (dtype, token grid): the ctypes structs plus
    the tensors they point to (kept alive here).  Rebuilt when a parameter changes."""

    def __init__(self, model, code: int, pos: torch.Tensor, embed_w: torch.Tensor, embed_b, embed_k: int):
        self.keep = []
        dev = pos.device

        def cw(t):  # GEMM matrix in compute dtype
            t = Fn.as_compute(t, code)
            self.keep.append(t)
            return t.data_ptr()

        def cf(t):  # fp32 vector (bias / LN affine); None -> NULL
            if t is None:
                return 0
            t = Fn.f32c(t)
            self.keep.append(t)
            return t.data_ptr()

        depth = len(model.blocks)
        self.blocks = (N.BlockWeights * max(depth, 1))()
        for i, blk in enumerate(model.blocks):
            b = self.blocks[i]
            b.ln1_w, b.ln1_b = cf(blk.norm1.weight), cf(blk.norm1.bias)
            b.qkv_w, b.qkv_b = cw(blk.attn.qkv.weight), cf(blk.attn.qkv.bias)
            b.proj_w, b.proj_b = cw(blk.attn.proj.weight), cf(blk.attn.proj.bias)
            b.ln2_w, b.ln2_b = cf(blk.norm2.weight), cf(blk.norm2.bias)
            b.fc1_w, b.fc1_b = cw(blk.mlp.fc1.weight), cf(blk.mlp.fc1.bias)
            b.fc2_w, b.fc2_b = cw(blk.mlp.fc2.weight), cf(blk.mlp.fc2.bias)
        w = N.VitWeights()
        w.dtype, w.dim, w.depth = code, model.embed_dim, depth
        w.heads = model.blocks[0].attn.num_heads if depth else 1
        w.hidden = model.blocks[0].mlp.fc1.out_features if depth else 4 * model.embed_dim
        w.ntok, w.embed_k = pos.shape[1], embed_k
        w.ln_eps = float(model.norm.eps)
        w.attn_scale = float(model.blocks[0].attn.scale) if depth else 0.0  # qk_scale or head_dim ** -0.5 (vision_transformer.py:112)
        w.embed_w, w.embed_b = cw(embed_w.reshape(embed_w.shape[0], -1)), cf(embed_b)
        w.cls, w.pos = cf(model.cls_token.reshape(-1)), cf(pos.reshape(pos.shape[1], -1))
        w.norm_w, w.norm_b = cf(model.norm.weight), cf(model.norm.bias)
        w.blocks = C.cast(self.blocks, C.POINTER(N.BlockWeights))
        self.w = w
        self.device = dev
        self._prepack(depth, dev)

    def _prepack(self, depth: int, dev) -> None:
        """Ring-ordered images of the block matrices for the streaming kernels (include/hipt_abmil.h, *_pk): one extra
        copy of the bf16 weights, made once per set of weights (this object is rebuilt when a parameter changes)."""
        if dev.type != "cuda":
            return
        lib = N.lib()
        for what, field in ((N.PACK_QKV, "qkv_pk"), (N.PACK_PROJ, "proj_pk"), (N.PACK_MLP, "mlp_pk"), (N.PACK_QKV_ATT, "qkv_att_pk")):
            nbytes = lib.hipt_vit_packed_bytes(C.byref(self.w), what)
            if not nbytes:
                continue
            # (the image format travels with the fused-MLP image: hipt_block_weights.mlp_pk_fmt)
            fmt = lib.hipt_vit_mlp_pack_format(C.byref(self.w)) if what == N.PACK_MLP else 0
            for i in range(depth):
                img = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                if what == N.PACK_MLP:
                    self.blocks[i].mlp_pk_fmt = fmt  # (read by hipt_vit_pack_weights' twin at launch time, set before either)
                N.call("hipt_vit_pack_weights", C.byref(self.w), i, what, N.ptr(img), N.stream_ptr(dev))
                self.keep.append(img)
                setattr(self.blocks[i], field, img.data_ptr())

    @property
    def ref(self):
        return C.byref(self.w)


class _HipVitMixin:
    """Shared machinery of VisionTransformer / VisionTransformer4K."""

    def _init_native(self):
        self._compute_dtype = _default_dtype()
        self._packed = {}      # device -> (key, _PackedVit): one weight image per device (nn.DataParallel replicas share this dict)
        self._pos_cache = {}   # device -> (key, interpolated positional table)
        self._warned_grad = False

    def __getstate__(self):
        """The device-side images (ctypes structs + packed tensors) are caches: never pickled, never deep-copied."""
        d = self.__dict__.copy()
        d["_packed"], d["_pos_cache"] = {}, {}
        return d

    def set_compute_dtype(self, name: str):
        N.dtype_code(name)
        self._compute_dtype = "bf16" if name in ("bf16", "bfloat16") else "fp32"
        return self

    @property
    def compute_dtype(self) -> str:
        return self._compute_dtype

    def _tensors(self):
        """The weight tensors of this module tree.  A ``nn.DataParallel`` replica has no ``parameters()`` (they are plain
        attributes there, torch/nn/parallel/replicate.py; the reference wraps the model so whenever it sees more than one
        GPU, extract_features_fp.py:217-218): take them from ``_former_parameters``."""
        ps = list(self.parameters())
        if ps:
            return ps
        return [t for m in self.modules() for t in getattr(m, "_former_parameters", {}).values() if t is not None]

    @property
    def weight_device(self):
        """Device the weights live on; survives DataParallel replication (``next(self.parameters())`` does not)."""
        return self.pos_embed.device

    def _version_key(self):
        return tuple((p.data_ptr(), p._version) for p in self._tensors())

    def _check_inference_only(self):
        """The HIP forwards are inference kernels (the ViTs are frozen feature extractors, hipt_model_utils.py:55-57):
        what they would silently get wrong is refused, what they merely do not provide is said once."""
        if self.training:
            for m in self.modules():
                if isinstance(m, nn.Dropout) and m.p > 0:
                    raise RuntimeError("HIP ViT forward: dropout p > 0 in train() mode is not implemented (inference kernels); call .eval()")
                if isinstance(m, DropPath) and (m.drop_prob or 0.) > 0:
                    raise RuntimeError("HIP ViT forward: drop_path > 0 in train() mode is not implemented (inference kernels); call .eval()")
        if torch.is_grad_enabled() and not self._warned_grad and any(p.requires_grad for p in self.parameters()):
            import warnings
            warnings.warn("HIP ViT forward returns tensors without grad_fn: no gradient flows into the ViT weights "
                          "(the reference freezes them, hipt_model_utils.py:55-57)", stacklevel=3)
            self._warned_grad = True

    def _pos_for(self, ntok_patches: int, w: int, h: int) -> torch.Tensor:
        dev = self.pos_embed.device
        key = (ntok_patches, w, h, self.pos_embed.data_ptr(), self.pos_embed._version)
        hit = self._pos_cache.get(dev)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, self._interpolate(ntok_patches, w, h).detach().float().contiguous())
            self._pos_cache[dev] = hit
        return hit[1]

    def _packed_for(self, pos: torch.Tensor) -> _PackedVit:
        self._check_inference_only()
        code = N.dtype_code(self._compute_dtype)
        key = (code, pos.data_ptr(), pos.shape[1], self._version_key())
        pk = self._packed.get(pos.device)
        if pk is None or pk[0] != key:
            ts = self._tensors()
            N.same_device(type(self).__name__, pos.device, *ts)
            ew, eb, ek = self._embed_params()
            pk = (key, _PackedVit(self, code, pos, ew, eb, ek))
            self._packed[pos.device] = pk
        return pk[1]

    # ---- shared tails -------------------------------------------------------------------
    def _blocks(self, pk, x, b0, b1, probs=None):
        nseq = x.shape[0]
        need = N.lib().hipt_vit_workspace_bytes(pk.ref, nseq)
        ws = Fn.workspace(x.device, need)
        N.call("hipt_vit_blocks", pk.ref, N.ptr(x), nseq, b0, b1, N.ptr(probs), N.ptr(ws), ws.numel(), N.stream_ptr(x.device))

    def _head(self, pk, x, cls_only: bool):
        nseq, ntok, D = x.shape
        out = torch.empty((nseq, D) if cls_only else (nseq, ntok, D), dtype=torch.float32, device=x.device)
        N.call("hipt_vit_head", pk.ref, N.ptr(x), nseq, 1 if cls_only else 0, N.ptr(out), N.stream_ptr(x.device))
        return out

    def get_last_selfattention(self, x):
        """[B, heads, N, N] softmax probabilities of the last block (vision_transformer.py:255-262)."""
        pk, tok = self._tokens(x)
        B, ntok, _ = tok.shape
        probs = torch.empty((B, pk.w.heads, ntok, ntok), dtype=torch.float32, device=tok.device)
        self._blocks(pk, tok, 0, pk.w.depth, probs)
        return probs

    def get_last_selfattention_cls(self, x):
        """[B, heads, N]: the [CLS] query's row of the last block's attention map, i.e. ``get_last_selfattention(x)[:, :, 0, :]``,
        which is all the heat-maps use (hipt_4k.py:143-158), from the one-query kernel: the [B, heads, N, N] tensor is never
        built (both ViTs, both compute dtypes)."""
        pk, tok = self._tokens(x)
        B, ntok, _ = tok.shape
        out = torch.empty((B, pk.w.heads, ntok), dtype=torch.float32, device=tok.device)
        need = N.lib().hipt_vit_workspace_bytes(pk.ref, B)
        ws = Fn.workspace(tok.device, need)
        N.call("hipt_vit_cls_attention", pk.ref, N.ptr(tok), B, N.ptr(out), N.ptr(ws), ws.numel(), N.stream_ptr(tok.device))
        return out

    def get_intermediate_layers(self, x, n=1):
        """Normed outputs of the ``n`` last blocks (vision_transformer.py:264-272)."""
        pk, tok = self._tokens(x)
        depth = pk.w.depth
        first = max(depth - n, 0)
        self._blocks(pk, tok, 0, first)
        out = []
        for i in range(first, depth):
            self._blocks(pk, tok, i, i + 1)
            out.append(self._head(pk, tok, cls_only=False))
        return out

    def prepare_tokens(self, x):
        return self._tokens(x)[1]


# ------------------------------------------------------------------------------------------
# Parameter-holding sub-modules.  Their own forward() exists for API compatibility and runs the
# same kernels through the fine-grained entry points.
# ------------------------------------------------------------------------------------------

def _code_of(module) -> int:
    return N.dtype_code(getattr(module, "_compute_dtype", _default_dtype()))


class Mlp(nn.Module):
    """fc1 -> GELU(erf) -> fc2 (vision_transformer.py:88-104)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        code = _code_of(self)
        h = Fn.linear(x, self.fc1.weight, self.fc1.bias, gelu=True, out_f32=False, dtype=code)
        return Fn.linear(h, self.fc2.weight, self.fc2.bias, dtype=code)


class Attention(nn.Module):
    """Multi-head self-attention (vision_transformer.py:107-131); forward returns (x, attn)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        code = _code_of(self)
        qkv = Fn.linear(x, self.qkv.weight, self.qkv.bias, out_f32=False, dtype=code)
        o, attn = Fn.attention(qkv, self.num_heads, self.scale, dtype=code, return_probs=True)
        return Fn.linear(o, self.proj.weight, self.proj.bias, dtype=code), attn


class DropPath(nn.Module):
    """Stochastic depth; identity on this (inference) path, kept for state/API parity."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        return x


class Block(nn.Module):
    """Pre-norm transformer block (vision_transformer.py:134-152)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, return_attention=False):
        code = _code_of(self)
        self.attn._compute_dtype = self.mlp._compute_dtype = "bf16" if code == N.HIPT_BF16 else "fp32"
        xn = Fn.layernorm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps,
                          out_dtype=code)
        qkv = Fn.linear(xn, self.attn.qkv.weight, self.attn.qkv.bias, out_f32=False, dtype=code)
        o, attn = Fn.attention(qkv, self.attn.num_heads, self.attn.scale, dtype=code, return_probs=return_attention)
        if return_attention:
            return attn
        x = Fn.linear(o, self.attn.proj.weight, self.attn.proj.bias, resid=x, dtype=code)
        xn = Fn.layernorm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, out_dtype=code)
        h = Fn.linear(xn, self.mlp.fc1.weight, self.mlp.fc1.bias, gelu=True, out_f32=False, dtype=code)
        return Fn.linear(h, self.mlp.fc2.weight, self.mlp.fc2.bias, resid=x, dtype=code)


class PatchEmbed(nn.Module):
    """Conv2d(k=s=patch) patch projection (vision_transformer.py:155-170)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        # im2col as a strided view (index plumbing), projection on the MFMA GEMM
        B, Cc, W, H = x.shape
        ps = self.patch_size
        cols = x.float().reshape(B, Cc, W // ps, ps, H // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, -1, Cc * ps * ps)
        return Fn.linear(cols, self.proj.weight.reshape(self.proj.out_channels, -1), self.proj.bias,
                         dtype=_code_of(self))


class VisionTransformer(_HipVitMixin, nn.Module):
    """ViT over 16x16-pixel tokens (vision_transformer.py:173-272), HIP-executed."""

    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size[0], patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        rates = torch.linspace(0, drop_path_rate, depth).tolist()
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=rates[i], norm_layer=norm_layer)
            for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self._init_native()

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # -- positional table: the reference's exact call (scale_factor form!), input independent, cached
    def _interpolate(self, npatch, w, h):
        Np = self.pos_embed.shape[1] - 1
        if npatch == Np and w == h:
            return self.pos_embed
        dim = self.pos_embed.shape[-1]
        ps = self.patch_embed.patch_size
        w0, h0 = w // ps + 0.1, h // ps + 0.1
        side = int(math.sqrt(Np))
        grid = self.pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
        grid = nn.functional.interpolate(grid, scale_factor=(w0 / math.sqrt(Np), h0 / math.sqrt(Np)), mode='bicubic')
        assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
        return torch.cat((self.pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def interpolate_pos_encoding(self, x, w, h):
        return self._interpolate(x.shape[1] - 1, w, h)

    def _embed_params(self):
        return self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.patch_embed.proj.weight[0].numel()

    def _layout(self, x):
        B, Cc, w, h = x.shape
        lay = N.ImageLayout(1, 1, w, h, h, w * h, Cc * w * h)
        return lay

    def _prep_input(self, x):
        N.require_cuda(x, type(self).__name__)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [B,3,W,H] images, got {tuple(x.shape)}")
        N.same_device(type(self).__name__, self.pos_embed.device, x)
        ps = self.patch_embed.patch_size
        if ps != 16:
            raise NotImplementedError("the HIP patch embedding is built for 16x16-pixel tokens")
        B, _, w, h = x.shape
        if w % ps or h % ps:
            # Conv2d(stride=ps) drops the remainder rows/cols
            x = x[:, :, : w - w % ps, : h - h % ps]
        x = x.detach().float().contiguous()
        npatch = (x.shape[2] // ps) * (x.shape[3] // ps)
        pos = self._pos_for(npatch, w, h)
        return x, self._packed_for(pos)

    def _tokens(self, x):
        x, pk = self._prep_input(x)
        B = x.shape[0]
        tok = torch.empty((B, pk.w.ntok, pk.w.dim), dtype=torch.float32, device=x.device)
        need = max(N.lib().hipt_vit_workspace_bytes(pk.ref, B), x.numel() * 2 + 256)
        ws = Fn.workspace(x.device, need)
        lay = self._layout(x)
        N.call("hipt_vit256_prepare_tokens", pk.ref, N.ptr(x), C.byref(lay), 0, B, N.ptr(tok), N.ptr(ws), ws.numel(),
               N.stream_ptr(x.device))
        return pk, tok

    def forward_features(self, x, layout=None, nseq=None, chunk=0):
        """[nseq, D] CLS features; ``layout`` lets HIPT_4K address 256x256 patches inside a region."""
        x, pk = (x, None) if layout is not None else self._prep_input(x)
        if layout is None:
            layout, nseq = self._layout(x), x.shape[0]
        else:
            N.require_cuda(x, type(self).__name__)
            N.same_device(type(self).__name__, self.pos_embed.device, x)
            pk = self._packed_for(self._pos_for((layout.patch_h // 16) * (layout.patch_w // 16), layout.patch_h,
                                                layout.patch_w))
        need = N.lib().hipt_vit256_forward_workspace_bytes(pk.ref, C.byref(layout), nseq, chunk)
        out = torch.empty((nseq, pk.w.dim), dtype=torch.float32, device=x.device)
        ws = Fn.workspace(x.device, need)
        N.call("hipt_vit256_forward", pk.ref, N.ptr(x), C.byref(layout), nseq, chunk, N.ptr(out), N.ptr(ws), ws.numel(),
               N.stream_ptr(x.device))
        return out

    def forward(self, x):
        return self.forward_features(x)


def vit_tiny(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                             qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
