"""CPU ORACLE, PyTorch-CPU form (test infrastructure, NOT product code).

The same restatement of the reference's hot path as ``oracle/hipt_oracle.py`` (numpy), written with PyTorch CPU
operators: this is the form SURVEY.md §8(d) / BASELINE.md §4 name as the timed CPU baseline ("the build's own PyTorch-CPU
restatement ... ``torch.set_num_threads(cores)`` ... warm-up 2 + min-of-5"), because the reference itself runs on these
very operators (``nn.Linear`` -> ``addmm``, ``nn.LayerNorm``, ``nn.GELU`` (erf), ``softmax``, ``matmul``; the reference's
Python cannot travel to the GPU box).  Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import it.

PARITY PIN: ``tests/test_oracle_vs_golden.py`` checks every function here against the fixtures captured from the
reference's own modules (``tests/golden/make_golden.py``).  Each function cites the reference lines it follows; ``p`` is
a dict of CPU tensors keyed by the reference's state-dict names; the positional table is taken from the numpy oracle's
``interpolate_pos_encoding`` (input independent, computed once -- exactly as the product does).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # partial(nn.LayerNorm, eps=1e-6): vision_transformer.py:282-286, vision_transformer4k.py:267-272


def to_torch(params_np):
    return {k: torch.from_numpy(v) for k, v in params_np.items()}


def _depth(p):
    return 1 + max(int(k.split(".")[1]) for k in p if k.startswith("blocks."))


def block(x, p, i, num_heads):
    """Block.forward (vision_transformer.py:146-152) with Attention (:119-131) and Mlp (:98-104)."""
    pre = f"blocks.{i}."
    B, N, C = x.shape
    dh = C // num_heads
    y = F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"], LN_EPS)
    qkv = F.linear(y, p[pre + "attn.qkv.weight"], p[pre + "attn.qkv.bias"]).reshape(B, N, 3, num_heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * (dh ** -0.5)).softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(y, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
    y = F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"], LN_EPS)
    y = F.gelu(F.linear(y, p[pre + "mlp.fc1.weight"], p[pre + "mlp.fc1.bias"]))
    return x + F.linear(y, p[pre + "mlp.fc2.weight"], p[pre + "mlp.fc2.bias"])


def _blocks_norm_cls(t, p, num_heads):
    for i in range(_depth(p)):
        t = block(t, p, i, num_heads)
    return F.layer_norm(t, (t.shape[-1],), p["norm.weight"], p["norm.bias"], LN_EPS)[:, 0]


def vit256_forward(x, p, pos, num_heads=6, patch_size=16):
    """VisionTransformer.forward (vision_transformer.py:235-253); ``pos`` = interpolated table [1, 1+n, D]."""
    B = x.shape[0]
    t = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=patch_size).flatten(2).transpose(1, 2)  # :167-170
    t = torch.cat([p["cls_token"].expand(B, -1, -1), t], dim=1) + pos
    return _blocks_norm_cls(t, p, num_heads)


def vit4k_forward(x, p, pos, num_heads=6):
    """VisionTransformer4K.forward (vision_transformer4k.py:223-246): x [B, 384, w, h]."""
    B = x.shape[0]
    t = F.gelu(F.linear(x.flatten(2, 3).transpose(1, 2), p["phi.0.weight"], p["phi.0.bias"]))
    t = torch.cat([p["cls_token"].expand(B, -1, -1), t], dim=1) + pos
    return _blocks_norm_cls(t, p, num_heads)


def hipt4k_forward(x, p256, p4k, pos256, pos4k):
    """HIPT_4K.forward (hipt_4k.py:48-76) for a region already a multiple of 256 on both sides."""
    b, c, W, H = x.shape
    w_256, h_256 = W // 256, H // 256
    batch = x.unfold(2, 256, 256).unfold(3, 256, 256).permute(0, 2, 3, 1, 4, 5).reshape(-1, c, 256, 256)  # :64-65
    f256 = torch.cat([vit256_forward(batch[i:i + 256], p256, pos256) for i in range(0, batch.shape[0], 256)], dim=0)  # :68-72
    grid = f256.reshape(w_256, h_256, 384).transpose(0, 1).transpose(0, 2).unsqueeze(0)  # :73
    return vit4k_forward(grid, p4k, pos4k)


def clam_sb_forward(h, p):
    """CLAM_SB.forward, eval path (model_clam.py:147-183) -> (logits, Y_prob, Y_hat, A_raw, M)."""
    g = next(int(k.split(".")[1]) for k in p if ".attention_a." in k)
    pre = f"attention_net.{g}."
    h1 = F.relu(F.linear(h, p["attention_net.0.weight"], p["attention_net.0.bias"]))
    a = torch.tanh(F.linear(h1, p[pre + "attention_a.0.weight"], p[pre + "attention_a.0.bias"]))
    b = torch.sigmoid(F.linear(h1, p[pre + "attention_b.0.weight"], p[pre + "attention_b.0.bias"]))
    A_raw = F.linear(a * b, p[pre + "attention_c.weight"], p[pre + "attention_c.bias"]).transpose(1, 0)
    M = torch.mm(F.softmax(A_raw, dim=1), h1)
    logits = F.linear(M, p["classifiers.weight"], p["classifiers.bias"])
    return logits, F.softmax(logits, dim=1), torch.topk(logits, 1, dim=1)[1], A_raw, M


# --------------------------------------------------------------------------------------
# CLAM training step: forward with autograd, so that tests can take gradients of any size / seed
# --------------------------------------------------------------------------------------

def clam_forward_train(h, p, n_classes, multi=False, k_sample=8, label=None, instance_eval=False, subtyping=False,
                       masks=None, instance_loss_fn=None):
    """CLAM_SB.forward / CLAM_MB.forward (model_clam.py:147-191, 226-264) as differentiable torch ops on ``p`` (tensors
    that may require grad).  ``masks`` = (m1 [N,S1], ma [N,S2], mb [N,S2]) scaled dropout masks or None (nn.Dropout after
    the ReLU :86 and inside Attn_Net_Gated :48-52).  Returns (logits, Y_prob, Y_hat, A_raw, dict)."""
    g = next(int(k.split(".")[1]) for k in p if ".attention_a." in k)
    pre = f"attention_net.{g}."
    h1 = F.relu(F.linear(h, p["attention_net.0.weight"], p["attention_net.0.bias"]))
    if masks is not None:
        h1 = h1 * masks[0]
    a = torch.tanh(F.linear(h1, p[pre + "attention_a.0.weight"], p[pre + "attention_a.0.bias"]))
    b = torch.sigmoid(F.linear(h1, p[pre + "attention_b.0.weight"], p[pre + "attention_b.0.bias"]))
    if masks is not None:
        a, b = a * masks[1], b * masks[2]
    A_raw = F.linear(a * b, p[pre + "attention_c.weight"], p[pre + "attention_c.bias"]).transpose(1, 0)  # [K, N]
    A = F.softmax(A_raw, dim=1)
    res = {}
    if instance_eval:
        loss_fn = instance_loss_fn or F.cross_entropy
        total, preds, targets = 0.0, [], []
        for c in range(n_classes):
            w, bb = p[f"instance_classifiers.{c}.weight"], p[f"instance_classifiers.{c}.bias"]
            Ab = (A[c] if multi else A[-1]).view(1, -1)
            if int(label) == c:  # inst_eval (:116-132)
                ids = torch.cat([torch.topk(Ab, k_sample)[1][-1], torch.topk(-Ab, k_sample, dim=1)[1][-1]])
                tg = torch.cat([torch.ones(k_sample, dtype=torch.long), torch.zeros(k_sample, dtype=torch.long)])
            elif subtyping:      # inst_eval_out (:135-145)
                ids, tg = torch.topk(Ab, k_sample)[1][-1], torch.zeros(k_sample, dtype=torch.long)
            else:
                continue
            lg = F.linear(h1[ids], w, bb)
            total = total + loss_fn(lg, tg)
            preds.append(lg.argmax(dim=1))
            targets.append(tg)
        if subtyping:
            total = total / n_classes
        res = dict(instance_loss=total, inst_preds=torch.cat(preds), inst_labels=torch.cat(targets))
    M = torch.mm(A, h1)
    if multi:
        logits = torch.stack([F.linear(M[c], p[f"classifiers.{c}.weight"], p[f"classifiers.{c}.bias"]).reshape(()) for c in range(n_classes)]).view(1, -1)
    else:
        logits = F.linear(M, p["classifiers.weight"], p["classifiers.bias"])
    res["features"] = M
    return logits, F.softmax(logits, dim=1), torch.topk(logits, 1, dim=1)[1], A_raw, res


def clam_train_step(h, params_np, label, n_classes=2, multi=False, k_sample=8, instance_eval=True, subtyping=False, bag_weight=0.7,
                    masks=None):
    """One step of train_loop_clam / train_loop (utils/core_utils.py:300-348, 373-426) -> (outputs dict, grads dict)."""
    p = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in params_np.items()}
    h = h.clone().requires_grad_(True)
    logits, y_prob, y_hat, a_raw, res = clam_forward_train(h, p, n_classes, multi, k_sample, label, instance_eval, subtyping, masks)
    loss = F.cross_entropy(logits, torch.tensor([label]))
    total = bag_weight * loss + (1 - bag_weight) * res["instance_loss"] if instance_eval else loss
    total.backward()
    out = dict(logits=logits.detach(), A_raw=a_raw.detach(), M=res["features"].detach(), loss=total.detach())
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    grads["bag"] = h.grad
    if instance_eval:
        out["inst_preds"], out["instance_loss"] = res["inst_preds"], res["instance_loss"].detach()
    return out, grads
