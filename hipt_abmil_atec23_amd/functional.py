"""Fine-grained operators of the HIP library as torch-tensor functions.

These are the units the parity tests exercise and what the sub-modules
(``Attention``, ``Mlp``, ``PatchEmbed`` ...) call when used on their own; the whole-model
forwards go through the coarse entry points (one C call per ViT / region / bag).
"""
from __future__ import annotations

import torch

from . import _native as N

_TORCH_DT = {N.HIPT_F32: torch.float32, N.HIPT_BF16: torch.bfloat16}
_workspaces = {}


def torch_dtype(code: int):
    return _TORCH_DT[code]


def workspace(device, nbytes: int, slot=0, zero: bool = False) -> torch.Tensor:
    """Grow-only per-device scratch (the library itself never allocates).  ``slot``: one scratch per concurrent use (a
    stream handle, a part index).  ``zero``: allocate zero-filled -- for scratch that carries the CLAM ticket block
    (include/hipt_abmil.h: zero before the first call, left zero by every call)."""
    key = (device.type, device.index, slot)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        n = max(int(nbytes), 1 << 20)
        ws = torch.zeros(n, dtype=torch.uint8, device=device) if zero else torch.empty(n, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def as_compute(t: torch.Tensor, code: int) -> torch.Tensor:
    """Contiguous copy/view of ``t`` in the compute dtype (weights and GEMM inputs)."""
    return t.detach().to(_TORCH_DT[code]).contiguous()


def f32c(t):
    return None if t is None else t.detach().float().contiguous()


def layernorm(x: torch.Tensor, weight, bias, eps: float = 1e-6, out_dtype: int = N.HIPT_F32) -> torch.Tensor:
    """nn.LayerNorm over the last dim (vision_transformer.py:138,142,195)."""
    N.require_cuda(x, "layernorm")
    x2 = x.detach().float().contiguous().view(-1, x.shape[-1])
    out = torch.empty(x2.shape, dtype=_TORCH_DT[out_dtype], device=x.device)
    N.call("hipt_layernorm", N.ptr(x2), x2.shape[1], N.ptr(f32c(weight)), N.ptr(f32c(bias)), N.ptr(out), out_dtype,
           x2.shape[1], x2.shape[0], x2.shape[1], float(eps), N.stream_ptr(x.device))
    return out.view(x.shape)


def linear(a: torch.Tensor, weight: torch.Tensor, bias=None, resid=None, gelu=False, relu=False,
           out_f32=True, dtype: int = N.HIPT_F32) -> torch.Tensor:
    """``epilogue(a @ weight.T + bias)`` (nn.Linear, vision_transformer.py:93-95,114,116)."""
    N.require_cuda(a, "linear")
    a2 = as_compute(a, dtype).view(-1, a.shape[-1])
    w = as_compute(weight, dtype)
    M, K = a2.shape
    Nn = w.shape[0]
    flags = (N.EPI_GELU if gelu else 0) | (N.EPI_RELU if relu else 0) | (N.EPI_OUT_F32 if out_f32 else 0)
    r = None
    if resid is not None:
        r = resid.detach().float().contiguous().view(M, Nn)
        flags |= N.EPI_RESID
    out = torch.empty((M, Nn), dtype=torch.float32 if out_f32 else _TORCH_DT[dtype], device=a.device)
    b = f32c(bias)
    N.call("hipt_linear", N.ptr(a2), K, N.ptr(w), K, N.ptr(b), N.ptr(r), N.ptr(out), Nn, M, Nn, K, dtype, flags,
           N.stream_ptr(a.device))
    return out.view(*a.shape[:-1], Nn)


def attention(qkv: torch.Tensor, num_heads: int, scale: float, dtype: int = N.HIPT_F32, return_probs: bool = False):
    """softmax(q k^T * scale) v from the fused qkv projection output [B, N, 3*C]
    (vision_transformer.py:122-128).  Returns (out [B,N,C], probs [B,H,N,N] or None)."""
    N.require_cuda(qkv, "attention")
    B, ntok, C3 = qkv.shape
    Cc = C3 // 3
    q = as_compute(qkv, dtype)
    out = torch.empty((B, ntok, Cc), dtype=_TORCH_DT[dtype], device=qkv.device)
    probs = torch.empty((B, num_heads, ntok, ntok), dtype=torch.float32, device=qkv.device) if return_probs else None
    N.call("hipt_attention", N.ptr(q), N.ptr(out), N.ptr(probs), B, ntok, num_heads, Cc // num_heads, float(scale), dtype,
           N.stream_ptr(qkv.device))
    return out, probs
