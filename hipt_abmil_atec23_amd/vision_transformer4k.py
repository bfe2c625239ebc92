"""ViT-4K host mirror (reference: ``HIPT_4K/vision_transformer4k.py``): a ViT over the
[w_256 x h_256] grid of ViT-256 [CLS] features.  Same constructor, attributes and state-dict keys
(``phi.0.*``, ``cls_token``, ``pos_embed``, ``blocks.*``, ``norm.*``); forward executed by the HIP
library (see vision_transformer.py in this package for the shared machinery)."""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from . import _native as N
from . import functional as Fn
from .vision_transformer import Attention, Block, DropPath, Mlp, _HipVitMixin, trunc_normal_  # noqa: F401


class VisionTransformer4K(_HipVitMixin, nn.Module):
    """vision_transformer4k.py:161-265."""

    def __init__(self, num_classes=0, img_size=[224], input_embed_dim=384, output_embed_dim=192,
                 depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, num_prototypes=64,
                 **kwargs):
        super().__init__()
        embed_dim = output_embed_dim
        self.num_features = self.embed_dim = embed_dim
        self.phi = nn.Sequential(nn.Linear(input_embed_dim, output_embed_dim), nn.GELU(), nn.Dropout(p=drop_rate))
        num_patches = int(img_size[0] // 16) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
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

    # positional table: patch size is 1 at this level (vision_transformer4k.py:208-209)
    def _interpolate(self, npatch, w, h):
        Np = self.pos_embed.shape[1] - 1
        if npatch == Np and w == h:
            return self.pos_embed
        dim = self.pos_embed.shape[-1]
        w0, h0 = w + 0.1, h + 0.1
        side = int(math.sqrt(Np))
        grid = self.pos_embed[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
        grid = nn.functional.interpolate(grid, scale_factor=(w0 / math.sqrt(Np), h0 / math.sqrt(Np)), mode='bicubic')
        assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
        return torch.cat((self.pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def interpolate_pos_encoding(self, x, w, h):
        return self._interpolate(x.shape[1] - 1, w, h)

    def _embed_params(self):
        return self.phi[0].weight, self.phi[0].bias, self.phi[0].in_features

    def _prep_input(self, x):
        N.require_cuda(x, type(self).__name__)
        if x.dim() != 4 or x.shape[1] != self.phi[0].in_features:
            raise ValueError(f"expected [B,{self.phi[0].in_features},w,h] feature grids, got {tuple(x.shape)}")
        N.same_device(type(self).__name__, self.pos_embed.device, x)
        self.mpp_feature = x  # the reference keeps the raw grid on the module (:225)
        B, E, w, h = x.shape
        tokens_in = x.detach().float().flatten(2, 3).transpose(1, 2).contiguous()  # [B, w*h, 384] (:227)
        return tokens_in, self._packed_for(self._pos_for(w * h, w, h))

    def _tokens(self, x):
        tokens_in, pk = self._prep_input(x)
        B = tokens_in.shape[0]
        tok = torch.empty((B, pk.w.ntok, pk.w.dim), dtype=torch.float32, device=x.device)
        ws = Fn.workspace(x.device, N.lib().hipt_vit_workspace_bytes(pk.ref, B))
        N.call("hipt_vit4k_prepare_tokens", pk.ref, N.ptr(tokens_in), B, N.ptr(tok), N.ptr(ws), ws.numel(),
               N.stream_ptr(x.device))
        return pk, tok

    def forward_tokens(self, tokens_in: torch.Tensor, w: int, h: int):
        """[B, w*h, 384] token-major features (what ViT-256 emits) -> [B, 192]; HIPT_4K's fast path."""
        N.require_cuda(tokens_in, type(self).__name__)
        N.same_device(type(self).__name__, self.pos_embed.device, tokens_in)
        pk = self._packed_for(self._pos_for(w * h, w, h))
        B = tokens_in.shape[0]
        need = N.lib().hipt_vit4k_forward_workspace_bytes(pk.ref, B)
        out = torch.empty((B, pk.w.dim), dtype=torch.float32, device=tokens_in.device)
        ws = Fn.workspace(tokens_in.device, need)
        N.call("hipt_vit4k_forward", pk.ref, N.ptr(tokens_in), B, N.ptr(out), N.ptr(ws), ws.numel(),
               N.stream_ptr(tokens_in.device))
        return out

    def forward(self, x):
        tokens_in, pk = self._prep_input(x)
        return self.forward_tokens(tokens_in, x.shape[2], x.shape[3])


def vit4k_xs(patch_size=16, **kwargs):
    return VisionTransformer4K(patch_size=patch_size, input_embed_dim=384, output_embed_dim=192, depth=6,
                               num_heads=6, mlp_ratio=4, qkv_bias=True,
                               norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
