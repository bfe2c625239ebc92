"""CLAM / ABMIL aggregator host mirror (reference: ``models/model_clam.py``).

``CLAM_SB`` and ``Attn_Net_Gated`` keep the reference's constructor arguments, attributes and
state-dict keys (``attention_net.0.*``, ``attention_net.{2|3}.attention_{a,b}.0.*``,
``...attention_c.*``, ``classifiers.*``, ``instance_classifiers.{c}.*``), so checkpoints written by
``utils/core_utils.py`` load with ``strict=True`` after the key cleaning of
``utils/eval_utils.py:51-57``.  Inference forwards (no autograd) run as ONE fused HIP pass over
the bag (csrc/abmil.hip).

Training: a forward that must be differentiated (``main.py`` trains this module,
``utils/core_utils.py:300-348, 373-426``) or that has active dropout runs the TRAINING kernels
(csrc/clam_train.hip) behind a ``torch.autograd.Function``: two launches forward (rows; softmax pooling + classifier +
on-device top-k of the instance branch), two backward (rows; weight gradients), fp32, for ``CLAM_SB`` and the K-branch
``CLAM_MB`` alike.  Only the pieces the caller supplies as Python callables stay PyTorch ops: the instance classifiers'
``Linear(S1, 2)`` on the 2k gathered rows and ``instance_loss_fn`` (``SmoothTop1SVM`` in the reference's scripts).
Outside the HIP path, by design: the ungated ``Attn_Net`` variant (``gate=False``; SURVEY.md allows the fallback) and
bags that live on the CPU -- training or not -- which run the module's PyTorch-op sequence on the CPU: the reference picks
the CPU itself where no GPU is visible (``relocate``, models/model_clam.py:102-106; SURVEY.md 8b "CPU tensors -> CPU
restatement"), so ``main.py`` / ``eval.py`` still run there.  That is a property of where the CALLER put the tensors, never a
fallback: a bag on a HIP device always takes the kernels, and a missing library raises (``NativeLibraryError``).  The ViTs
have no CPU path at all.
"""
from __future__ import annotations

import ctypes as C_
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _native as N
from . import functional as Fn

SIZE_DICT = {"tinier3": [1024, 32, 8], "256": [256, 64, 16], "tinier_resnet18": [512, 64, 16],
             "tinier2_resnet18": [512, 32, 8], "tiny_resnet18": [512, 128, 32], "small_resnet18": [512, 256, 64],
             "tinier": [1024, 64, 16], "tiny128": [1024, 128, 32], "tiny": [1024, 256, 64], "small": [1024, 512, 256],
             "big": [1024, 512, 384], "hipt_big": [192, 128, 64], "hipt_medium": [192, 64, 32],
             "hipt_small": [192, 32, 16], "hipt_smaller": [192, 16, 8], "hipt_smallest": [192, 8, 4],
             # added for 384-d ViT-256 [CLS] bags (BASELINE configs 1 and 4; SURVEY.md §8d "384 sizing")
             "hipt_384": [384, 128, 64]}


def initialize_weights(module):
    """xavier-normal Linear weights, zero bias (utils/utils.py:217-225)."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight)
            m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm1d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def _default_dtype() -> str:
    return os.environ.get("HIPT_AMD_DTYPE", "fp32")


def _needs_autograd(module, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    return any(t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


def _all_on_cpu(module, x) -> bool:
    """The caller keeps module AND input on the CPU (what the reference's relocate() does where no GPU is visible)."""
    return not x.is_cuda and not any(p.is_cuda for p in module.parameters())


class Attn_Net(nn.Module):
    """Ungated attention head (model_clam.py:15-31); plain PyTorch module (not on the HIP path)."""

    def __init__(self, L=1024, D=256, dropout=0.25, n_classes=1):
        super().__init__()
        layers = [nn.Linear(L, D), nn.Tanh()]
        if dropout > 0:
            layers.append(nn.Dropout(dropout))
        layers.append(nn.Linear(D, n_classes))
        self.module = nn.Sequential(*layers)

    def forward(self, x):
        return self.module(x), x


class Attn_Net_Gated(nn.Module):
    """Gated attention head (model_clam.py:41-64): A = (tanh(x Wa^T) * sigmoid(x Wb^T)) Wc^T."""

    def __init__(self, L=1024, D=256, dropout=0.0, n_classes=1):
        super().__init__()
        a = [nn.Linear(L, D), nn.Tanh()]
        b = [nn.Linear(L, D), nn.Sigmoid()]
        if dropout > 0:
            a.append(nn.Dropout(dropout))
            b.append(nn.Dropout(dropout))
        self.attention_a = nn.Sequential(*a)
        self.attention_b = nn.Sequential(*b)
        self.attention_c = nn.Linear(D, n_classes)
        self._compute_dtype = _default_dtype()
        self._packed = None

    def set_compute_dtype(self, name):
        N.dtype_code(name)
        self._compute_dtype = "bf16" if name in ("bf16", "bfloat16") else "fp32"
        return self

    def _torch_forward(self, x):
        return self.attention_c(self.attention_a(x).mul(self.attention_b(x))), x

    def __getstate__(self):  # the packed weight image (ctypes struct + tensors) is a cache: never pickled / deep-copied
        d = self.__dict__.copy()
        d["_packed"] = None
        return d

    def _pack(self, device):
        code = N.dtype_code(self._compute_dtype)
        N.same_device("Attn_Net_Gated", device, *self.parameters())
        key = (code, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if self._packed is None or self._packed[0] != key:
            wa, wb = self.attention_a[0], self.attention_b[0]
            keep = dict(
                wab=Fn.as_compute(torch.cat([wa.weight, wb.weight], dim=0), code),
                bab=Fn.f32c(torch.cat([wa.bias, wb.bias], dim=0)),
                wc=Fn.f32c(self.attention_c.weight.reshape(-1)), bc=Fn.f32c(self.attention_c.bias))
            w = N.ClamWeights()
            w.dtype, w.s0, w.s1, w.s2, w.n_classes = code, 0, wa.in_features, wa.out_features, 0
            w.wab, w.bab, w.wc, w.bc = (keep[k].data_ptr() for k in ("wab", "bab", "wc", "bc"))
            self._packed = (key, w, keep)
        return self._packed[1]

    def forward(self, x):
        dropout_on = self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.modules())
        if self.attention_c.out_features != 1 or dropout_on or _needs_autograd(self, x) or _all_on_cpu(self, x):
            return self._torch_forward(x)  # training / multi-branch (CLAM_MB) / a CPU tensor: PyTorch ops on the same device
        N.require_cuda(x, "Attn_Net_Gated")
        w = self._pack(x.device)
        if x.dim() < 1 or x.shape[-1] != w.s1:
            raise RuntimeError(f"Attn_Net_Gated: input {tuple(x.shape)} does not end in the L = {w.s1} features the module was built for")
        xin = Fn.as_compute(x, w.dtype).reshape(-1, w.s1)  # nn.Linear semantics: any leading dims
        n = xin.shape[0]
        A = torch.empty((n, 1), dtype=torch.float32, device=x.device)
        if n:
            need = n * 2 * w.s2 * 4 + 4096
            st = N.stream_ptr(x.device)
            ws = Fn.workspace(x.device, need, ("gated", st.value))
            N.call("hipt_attn_net_gated", C_.byref(w), N.ptr(xin), n, N.ptr(A), N.ptr(ws), ws.numel(), st)
        return A.reshape(*x.shape[:-1], 1), x


def _train_supported(sizes, k_att, n_classes, need_dbag=False) -> bool:
    """Do the training kernels take this shape (widths, branches, classes AND the LDS their row tiles need)?  Asked of the
    library itself (hipt_clam_train_shape_supported), so the answer cannot drift from the kernels' own checks."""
    if not (all(v > 0 and v % 4 == 0 for v in sizes) and 1 <= k_att <= 8 and 1 <= n_classes <= 8):
        return False
    return bool(N.lib().hipt_clam_train_shape_supported(int(sizes[0]), int(sizes[1]), int(sizes[2]), int(k_att), int(n_classes), int(need_dbag)))


class _ClamTrainFn(torch.autograd.Function):
    """Differentiable CLAM forward on the training kernels (include/hipt_abmil.h: hipt_clam_train_forward / _backward).

    forward(cfg, bag, m1, ma, mb, w1, b1, wa, ba, wb, bb, wc, bc, *cls) -> (logits [1,C], A_raw [K,N], M [K,S1],
    h1_sel [K,2,k,S1], topk_ids [K,2,k] int64, Y_hat [1,1] int64); ``cls`` = (weight [C,S1], bias [C]) for CLAM_SB or the
    2C tensors of CLAM_MB's per-class ``Linear(S1, 1)`` (w_0, b_0, w_1, b_1, ...)."""

    @staticmethod
    def forward(ctx, cfg, bag, m1, ma, mb, w1, b1, wa, ba, wb, bb, wc, bc, *cls):
        dev = bag.device
        K, C, multi, k_sel = cfg["K"], cfg["C"], cfg["multi"], cfg["k_sample"]
        f = lambda t: None if t is None else t.detach().float().contiguous()
        x = f(bag)
        n, S0 = x.shape
        S1, S2 = w1.shape[0], wa.shape[0]
        if multi:
            wcls = torch.cat([f(cls[2 * c]).reshape(1, S1) for c in range(C)], dim=0)
            bcls = torch.cat([f(cls[2 * c + 1]).reshape(1) for c in range(C)], dim=0)
        else:
            wcls, bcls = f(cls[0]), f(cls[1])
        keep = [x, f(m1), f(ma), f(mb), f(w1), f(b1), f(wa), f(ba), f(wb), f(bb), f(wc).reshape(K, S2), f(bc), wcls, bcls]
        N.same_device("CLAM training forward", dev, *keep)
        w = N.ClamTrainWeights()
        w.s0, w.s1, w.s2, w.n_att, w.n_classes, w.multi_branch = S0, S1, S2, K, C, int(multi)
        for name, t in zip(("w1", "b1", "wa", "ba", "wb", "bb", "wc", "bc", "wcls", "bcls"), keep[4:]):
            setattr(w, name, t.data_ptr())
        e = lambda *shape, dt=torch.float32: torch.empty(shape, dtype=dt, device=dev)
        h1, t, s, A_raw, stats, M = e(n, S1), e(n, S2), e(n, S2), e(K, n), e(K, 2), e(K, S1)
        logits, y_prob, y_hat = e(1, C), e(1, C), e(1, 1, dt=torch.int64)
        ids = e(K, 2, k_sel, dt=torch.int64) if k_sel > 0 else None
        h1_sel = e(K, 2, k_sel, S1) if k_sel > 0 else None
        st = N.stream_ptr(dev)
        N.call("hipt_clam_train_forward", C_.byref(w), N.ptr(x), n, N.ptr(keep[1]), N.ptr(keep[2]), N.ptr(keep[3]), N.ptr(h1), N.ptr(t),
               N.ptr(s), N.ptr(A_raw), N.ptr(stats), N.ptr(M), N.ptr(logits), N.ptr(y_prob), N.ptr(y_hat), k_sel, N.ptr(ids), N.ptr(h1_sel), st)
        ctx.cfg = cfg
        ctx.w = w          # (the pointers stay valid: every tensor they name is in ctx.keep)
        ctx.keep = keep
        # `f` returns the parameter's own storage when it is already fp32 and contiguous: an in-place update between forward
        # and backward (optimizer.step) would silently change what the backward reads -- autograd's saved-tensor version
        # check raises there, and so does this
        srcs = [bag, m1, ma, mb, w1, b1, wa, ba, wb, bb, wc, bc, *cls]
        ctx.versions = [(t, t._version) for t in srcs if t is not None]
        ctx.saved = (h1, t, s, A_raw, stats, M, ids)
        ctx.set_materialize_grads(False)
        if ids is None:
            ids, h1_sel = torch.empty((K, 2, 0), dtype=torch.int64, device=dev), torch.empty((K, 2, 0, S1), device=dev)
        ctx.mark_non_differentiable(ids, y_hat)
        return logits, A_raw, M, h1_sel, ids, y_hat

    @staticmethod
    def backward(ctx, dlogits, dA_raw, dM, dh1_sel, _ids, _yhat):
        cfg, w, keep = ctx.cfg, ctx.w, ctx.keep
        for t, v in ctx.versions:
            if t._version != v:
                raise RuntimeError("one of the variables needed for gradient computation has been modified by an inplace operation "
                                   f"(CLAM training step: a {tuple(t.shape)} tensor is at version {t._version}, expected {v})")
        x, m1, ma, mb = keep[:4]
        h1, t, s, A_raw, stats, M, ids = ctx.saved
        dev = x.device
        n, S0 = x.shape
        S1, S2, K, C = w.s1, w.s2, w.n_att, w.n_classes
        z = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g = N.ClamTrainGrads()
        out = dict(dw1=z(S1, S0), db1=z(S1), dwa=z(S2, S1), dba=z(S2), dwb=z(S2, S1), dbb=z(S2), dwc=z(K, S2), dbc=z(K), dwcls=z(C, S1), dbcls=z(C))
        if ctx.needs_input_grad[1]:
            out["dbag"] = z(n, S0)
        for name, tns in out.items():
            setattr(g, name, tns.data_ptr())
        c = lambda tns: None if tns is None else tns.detach().float().contiguous()
        dl = c(dlogits) if dlogits is not None else torch.zeros((1, C), device=dev)
        dA, dMx, dsel = c(dA_raw), c(dM), c(dh1_sel)
        n_sel = 0
        sel_ids = None
        if dsel is not None and ids is not None and dsel.numel():
            sel_ids, n_sel = ids.reshape(-1), ids.numel()
        st = N.stream_ptr(dev)
        need = N.lib().hipt_clam_train_workspace_bytes(C_.byref(w), n)
        ws = Fn.workspace(dev, need, ("clam_train", st.value))
        N.call("hipt_clam_train_backward", C_.byref(w), N.ptr(x), n, N.ptr(m1), N.ptr(ma), N.ptr(mb), N.ptr(h1), N.ptr(t), N.ptr(s), N.ptr(A_raw),
               N.ptr(stats), N.ptr(M), N.ptr(dl), N.ptr(dA), N.ptr(dMx), N.ptr(sel_ids), N.ptr(dsel if n_sel else None), n_sel, C_.byref(g),
               N.ptr(ws), ws.numel(), st)
        if cfg["multi"]:
            gcls = []
            for cl in range(C):
                gcls += [out["dwcls"][cl:cl + 1], out["dbcls"][cl:cl + 1]]
        else:
            gcls = [out["dwcls"], out["dbcls"]]
        wc_shape = cfg["wc_shape"]
        return (None, out.get("dbag"), None, None, None, out["dw1"], out["db1"], out["dwa"], out["dba"], out["dwb"], out["dbb"],
                out["dwc"].reshape(wc_shape), out["dbc"], *gcls)


class CLAM_SB(nn.Module):
    """Single-branch CLAM / ABMIL (model_clam.py:77-191).

    ``size_arg`` is a key of the reference's size table (plus ``'hipt_384'``) or an explicit
    ``[S0, S1, S2]`` list."""

    _multi = False

    def __init__(self, gate=True, size_arg="small", dropout=0.0, k_sample=8, n_classes=2,
                 instance_loss_fn=nn.CrossEntropyLoss(), subtyping=False):
        super().__init__()
        self.size_dict = dict(SIZE_DICT)
        size = list(size_arg) if isinstance(size_arg, (list, tuple)) else self.size_dict[size_arg]
        self._build(gate, size, dropout, k_sample, n_classes, instance_loss_fn, subtyping, att_branches=1,
                    classifiers=lambda: nn.Linear(size[1], n_classes))
        initialize_weights(self)

    def _build(self, gate, size, dropout, k_sample, n_classes, instance_loss_fn, subtyping, att_branches, classifiers):
        fc = [nn.Linear(size[0], size[1]), nn.ReLU()]
        if dropout > 0:
            fc.append(nn.Dropout(dropout))
        head = Attn_Net_Gated if gate else Attn_Net
        fc.append(head(L=size[1], D=size[2], dropout=dropout, n_classes=att_branches))
        self.attention_net = nn.Sequential(*fc)
        self.classifiers = classifiers()  # (registration order = the reference's: attention_net, classifiers, instance_classifiers)
        self.instance_classifiers = nn.ModuleList([nn.Linear(size[1], 2) for _ in range(n_classes)])
        self.k_sample = k_sample
        self.instance_loss_fn = instance_loss_fn
        self.n_classes = n_classes
        self.subtyping = subtyping
        self._gate = gate
        self._dropout = dropout
        self._sizes = tuple(size)
        self._compute_dtype = _default_dtype()
        self._packed = None

    def set_compute_dtype(self, name):
        N.dtype_code(name)
        self._compute_dtype = "bf16" if name in ("bf16", "bfloat16") else "fp32"
        return self

    @property
    def compute_dtype(self):
        return self._compute_dtype

    def relocate(self):
        device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.attention_net = self.attention_net.to(device)
        self.classifiers = self.classifiers.to(device)
        self.instance_classifiers = self.instance_classifiers.to(device)

    @staticmethod
    def create_positive_targets(length, device):
        return torch.full((length,), 1, device=device).long()

    @staticmethod
    def create_negative_targets(length, device):
        return torch.full((length,), 0, device=device).long()

    # ---- instance-level branches (model_clam.py:116-145) ------------------------------------------
    # rows_p / rows_n: the h1 rows of the k highest / k lowest attention scores of the branch
    def _inst_eval_rows(self, rows_p, rows_n, classifier):
        device = rows_p.device
        targets = torch.cat([self.create_positive_targets(self.k_sample, device),
                             self.create_negative_targets(self.k_sample, device)], dim=0)
        logits = classifier(torch.cat([rows_p, rows_n], dim=0))
        preds = torch.topk(logits, 1, dim=1)[1].squeeze(1)
        return self.instance_loss_fn(logits, targets), preds, targets

    def _inst_eval_out_rows(self, rows_p, classifier):
        targets = self.create_negative_targets(self.k_sample, rows_p.device)
        logits = classifier(rows_p)
        preds = torch.topk(logits, 1, dim=1)[1].squeeze(1)
        return self.instance_loss_fn(logits, targets), preds, targets

    def inst_eval(self, A, h, classifier):
        """Reference signature (model_clam.py:116-132): top-k of A by PyTorch ops on whatever device A lives on."""
        if A.dim() == 1:
            A = A.view(1, -1)
        top_p_ids = torch.topk(A, self.k_sample)[1][-1]
        top_n_ids = torch.topk(-A, self.k_sample, dim=1)[1][-1]
        return self._inst_eval_rows(torch.index_select(h, 0, top_p_ids), torch.index_select(h, 0, top_n_ids), classifier)

    def inst_eval_out(self, A, h, classifier):
        if A.dim() == 1:
            A = A.view(1, -1)
        top_p_ids = torch.topk(A, self.k_sample)[1][-1]
        return self._inst_eval_out_rows(torch.index_select(h, 0, top_p_ids), classifier)

    def _instance_branch(self, rows_of, label):
        """The instance loop of forward (:156-178 / :234-245).  rows_of(branch) -> (rows_p [k,S1], rows_n [k,S1])."""
        total, all_preds, all_targets = 0.0, [], []
        inst_labels = F.one_hot(label, num_classes=self.n_classes).squeeze()
        for i, classifier in enumerate(self.instance_classifiers):
            rows_p, rows_n = rows_of(i if self._multi else 0)
            if inst_labels[i].item() == 1:
                loss, preds, targets = self._inst_eval_rows(rows_p, rows_n, classifier)
            elif self.subtyping:
                loss, preds, targets = self._inst_eval_out_rows(rows_p, classifier)
            else:
                continue
            all_preds.extend(preds.cpu().numpy())
            all_targets.extend(targets.cpu().numpy())
            total += loss
        if self.subtyping:
            total /= len(self.instance_classifiers)
        return {'instance_loss': total, 'inst_labels': np.array(all_targets), 'inst_preds': np.array(all_preds)}

    def _bag_logits(self, M):
        return self.classifiers(M)  # :181

    # ---- the reference op sequence as PyTorch ops (ungated head; CPU tensors in training) ----------
    def _torch_forward(self, h, label, instance_eval, return_features, attention_only):
        A, h = self.attention_net(h)
        A = torch.transpose(A, 1, 0)
        if attention_only:
            return A
        A_raw = A
        A = F.softmax(A, dim=1)
        results = {}
        if instance_eval:
            def rows_of(b):
                a = A[b].view(1, -1)
                return (torch.index_select(h, 0, torch.topk(a, self.k_sample)[1][-1]),
                        torch.index_select(h, 0, torch.topk(-a, self.k_sample, dim=1)[1][-1]))
            results = self._instance_branch(rows_of, label)
        M = torch.mm(A, h)
        logits = self._bag_logits(M)
        Y_hat = torch.topk(logits, 1, dim=1)[1]
        Y_prob = F.softmax(logits, dim=1)
        if return_features:
            results.update({'features': M})
        return logits, Y_prob, Y_hat, A_raw, results

    # ---- HIP training path ----------------------------------------------------------------------------
    def _cls_tensors(self):
        return [self.classifiers.weight, self.classifiers.bias]

    def _train_forward(self, h, label, instance_eval, return_features, attention_only):
        if h.dim() != 2 or h.shape[0] == 0 or h.shape[1] != self._sizes[0]:
            raise ValueError(f"expected a non-empty [N, {self._sizes[0]}] bag, got {tuple(h.shape)}")
        fc1, gated = self.attention_net[0], self.attention_net[-1]
        wa, wb, wc = gated.attention_a[0], gated.attention_b[0], gated.attention_c
        n, (S0, S1, S2), K = h.shape[0], self._sizes, wc.out_features
        m1 = ma = mb = None
        if self.training and self._dropout > 0:  # the masks nn.Dropout would draw, in the reference's order (:86, :48-52)
            ones = lambda c: torch.ones((n, c), dtype=torch.float32, device=h.device)
            m1, ma, mb = (F.dropout(ones(S1), self._dropout, True), F.dropout(ones(S2), self._dropout, True),
                          F.dropout(ones(S2), self._dropout, True))
        k_sel = self.k_sample if (instance_eval and not attention_only) else 0
        cfg = dict(K=K, C=self.n_classes, multi=self._multi, k_sample=k_sel, wc_shape=tuple(wc.weight.shape))
        logits, A_raw, M, h1_sel, _ids, Y_hat = _ClamTrainFn.apply(
            cfg, h, m1, ma, mb, fc1.weight, fc1.bias, wa.weight, wa.bias, wb.weight, wb.bias, wc.weight, wc.bias, *self._cls_tensors())
        if attention_only:
            return A_raw
        results = self._instance_branch(lambda b: (h1_sel[b, 0], h1_sel[b, 1]), label) if instance_eval else {}
        Y_prob = F.softmax(logits, dim=1)
        if return_features:
            results.update({'features': M})
        return logits, Y_prob, Y_hat, A_raw, results

    def _use_train_kernels(self, h, dropout_on) -> bool:
        if not self._gate or not h.is_cuda:
            return False
        K = self.attention_net[-1].attention_c.out_features
        if not _train_supported(self._sizes, K, self.n_classes, need_dbag=torch.is_grad_enabled() and h.requires_grad):
            return False
        return dropout_on or _needs_autograd(self, h)

    # ---- HIP inference path ---------------------------------------------------------------------------
    def __getstate__(self):  # the packed weight image (ctypes struct + tensors) is a cache: never pickled / deep-copied
        d = self.__dict__.copy()
        d["_packed"] = None
        return d

    def _pack(self, device):
        code = N.dtype_code(self._compute_dtype)
        N.same_device(type(self).__name__, device, *self.parameters())  # e.g. relocate() never called: a clean error, not a GPU fault
        key = (code, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if self._packed is None or self._packed[0] != key:
            fc1, gated = self.attention_net[0], self.attention_net[-1]
            wa, wb, wc = gated.attention_a[0], gated.attention_b[0], gated.attention_c
            keep = dict(
                w1=Fn.as_compute(fc1.weight, code), b1=Fn.f32c(fc1.bias),
                wab=Fn.as_compute(torch.cat([wa.weight, wb.weight], dim=0), code),
                bab=Fn.f32c(torch.cat([wa.bias, wb.bias], dim=0)),
                wc=Fn.f32c(wc.weight.reshape(-1)), bc=Fn.f32c(wc.bias),
                wcls=Fn.f32c(self.classifiers.weight), bcls=Fn.f32c(self.classifiers.bias))
            w = N.ClamWeights()
            w.dtype, w.s0, w.s1, w.s2 = code, fc1.in_features, fc1.out_features, wa.out_features
            w.n_classes = self.classifiers.out_features
            for k, t in keep.items():
                setattr(w, k, t.data_ptr())
            # |A_raw - bc| <= sum |wc_j| (tanh * sigmoid lies in (-1, 1)): lets the streaming kernel exponentiate against a fixed
            # shift instead of a running maximum (include/hipt_abmil.h, hipt_clam_weights.logit_bound); one tiny reduction per
            # set of weights, read back here once
            # (0 means "unknown" in the ABI: an all-zero attention_c -- a zero-initialised head -- has the bound 0 and still takes the
            #  streaming kernel through the smallest positive bound.  The read-back synchronises once per set of weights; pack outside
            #  a graph capture.)
            w.logit_bound = max(float(keep["wc"].abs().sum().item()), 1e-30)
            # the streaming kernel's LDS image of these weights (bf16 [384|192,128,64]): packed once here, copied straight by
            # LDS-DMA at every launch
            nb = N.lib().hipt_clam_stream_packed_bytes(C_.byref(w))
            if nb:
                keep["stream_pk"] = torch.empty(nb, dtype=torch.uint8, device=device)
                N.call("hipt_clam_stream_pack", C_.byref(w), N.ptr(keep["stream_pk"]), N.stream_ptr(device))
                w.stream_pk = keep["stream_pk"].data_ptr()
            self._packed = (key, w, keep)
        return self._packed[1]

    def forward(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        dropout_on = self.training and self._dropout > 0
        if self._use_train_kernels(h, dropout_on):
            N.same_device(type(self).__name__, h.device, *self.parameters())
            return self._train_forward(h, label, instance_eval, return_features, attention_only)
        # whatever the training kernels do not take (ungated head, CPU tensors, > 8 classes, widths beyond their LDS) keeps the
        # PyTorch-op sequence as soon as dropout is active or a gradient is needed: the inference kernels have neither
        # ... and so does a bag the caller keeps on the CPU (the reference's relocate() chooses the CPU where there is no GPU,
        # models/model_clam.py:102-106): PyTorch ops on the CPU.  A bag on a HIP device never comes this way.
        # (a CPU bag handed to a module that lives on a HIP device is a caller's mistake and raises below)
        if not self._gate or dropout_on or _needs_autograd(self, h) or _all_on_cpu(self, h):
            return self._torch_forward(h, label, instance_eval, return_features, attention_only)
        if self._multi:
            return self._multi_infer(h, label, instance_eval, return_features, attention_only)
        N.require_cuda(h, "CLAM_SB")
        if h.dim() != 2 or h.shape[0] == 0:
            raise ValueError(f"expected a non-empty [N, {self.attention_net[0].in_features}] bag, got {tuple(h.shape)}")
        w = self._pack(h.device)
        if h.shape[1] != w.s0:
            raise ValueError(f"bag width {h.shape[1]} != model input width {w.s0}")
        dev = h.device
        bag = Fn.as_compute(h, w.dtype)
        n = bag.shape[0]
        A_raw = torch.empty((1, n), dtype=torch.float32, device=dev)
        st = N.stream_ptr(dev)
        # one scratch (partials + finish ticket) per stream: two CLAM calls on two streams must not share them
        ws = Fn.workspace(dev, N.lib().hipt_clam_workspace_bytes(C_.byref(w), n), ("clam", st.value), zero=True)
        if attention_only:
            N.call("hipt_clam_sb_forward", C_.byref(w), N.ptr(bag), n, 1, N.ptr(A_raw), None, None, None, None,
                   N.ptr(ws), ws.numel(), st)
            return A_raw
        M = torch.empty((1, w.s1), dtype=torch.float32, device=dev)
        logits = torch.empty((1, w.n_classes), dtype=torch.float32, device=dev)
        Y_prob = torch.empty_like(logits)
        Y_hat = torch.empty((1, 1), dtype=torch.int64, device=dev)
        N.call("hipt_clam_sb_forward", C_.byref(w), N.ptr(bag), n, 0, N.ptr(A_raw), N.ptr(M), N.ptr(logits), N.ptr(Y_prob),
               N.ptr(Y_hat), N.ptr(ws), ws.numel(), st)
        results = {}
        if instance_eval:
            # inst_eval in eval mode (validate_clam, core_utils.py:506-560): top-k ids on the device (softmax is monotone:
            # the ids of A_raw are those of softmax(A_raw)), then only the 2k selected rows of h1 are recomputed by the
            # library instead of materialising h1 [N,S1]
            if self.k_sample > n:
                raise RuntimeError(f"selected index k out of range: k_sample={self.k_sample} > {n} rows (torch.topk, model_clam.py:120)")
            ids = torch.empty((1, 2, self.k_sample), dtype=torch.int64, device=dev)
            N.call("hipt_topk_rows", N.ptr(A_raw), 1, n, self.k_sample, N.ptr(ids), st)
            rows = torch.empty((2 * self.k_sample, w.s1), dtype=torch.float32, device=dev)
            N.call("hipt_clam_gather_h1", C_.byref(w), N.ptr(bag), N.ptr(ids), 2 * self.k_sample, N.ptr(rows), st)
            results = self._instance_branch(lambda b: (rows[:self.k_sample], rows[self.k_sample:]), label)
        if return_features:
            results.update({'features': M})
        return logits, Y_prob, Y_hat, A_raw, results


class CLAM_MB(CLAM_SB):
    """Multi-branch CLAM (model_clam.py:193-264): one attention branch and one ``Linear(S1, 1)`` bag classifier per class.

    Inference (eval mode, no gradient asked for) runs the SAME streaming kernels as ``CLAM_SB`` -- MFMA projections, one pass
    over the bag per branch (``hipt_clam_sb_forward`` with the branch's ``attention_c`` row and its one-row classifier: the K
    branches share W1 and [Wa; Wb], only ``wc [K, S2]``, ``bc [K]`` and the pooled sums differ), in both compute dtypes; a
    forward that must be differentiable (or has active dropout) runs the K-branch fp32 training kernels of csrc/clam_train.hip;
    ``gate=False`` and CPU tensors take the PyTorch-op sequence."""

    _multi = True
    one_pass = True  # inference takes hipt_clam_mb_forward (one pass over the bag for all branches) where the library has it; False: branch by branch

    def __init__(self, gate=True, size_arg="small", dropout=0.0, k_sample=8, n_classes=2,
                 instance_loss_fn=nn.CrossEntropyLoss(), subtyping=False):
        nn.Module.__init__(self)
        self.size_dict = dict(SIZE_DICT)
        size = list(size_arg) if isinstance(size_arg, (list, tuple)) else self.size_dict[size_arg]
        self._build(gate, size, dropout, k_sample, n_classes, instance_loss_fn, subtyping, att_branches=n_classes,
                    classifiers=lambda: nn.ModuleList([nn.Linear(size[1], 1) for _ in range(n_classes)]))
        initialize_weights(self)

    def _cls_tensors(self):
        out = []
        for c in self.classifiers:
            out += [c.weight, c.bias]
        return out

    def _pack_branches(self, device):
        """One ``hipt_clam_weights`` per attention branch for the inference kernels: shared W1 / [Wa; Wb] tensors, the branch's row of
        ``attention_c`` and its ``Linear(S1, 1)`` as a one-class bag classifier (cached like ``_pack``)."""
        code = N.dtype_code(self._compute_dtype)
        N.same_device(type(self).__name__, device, *self.parameters())
        key = ("mb", code, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        if self._packed is None or self._packed[0] != key:
            fc1, gated = self.attention_net[0], self.attention_net[-1]
            wa, wb, wc = gated.attention_a[0], gated.attention_b[0], gated.attention_c
            shared = dict(
                w1=Fn.as_compute(fc1.weight, code), b1=Fn.f32c(fc1.bias),
                wab=Fn.as_compute(torch.cat([wa.weight, wb.weight], dim=0), code),
                bab=Fn.f32c(torch.cat([wa.bias, wb.bias], dim=0)))
            wc_all, bc_all = Fn.f32c(wc.weight), Fn.f32c(wc.bias)
            bounds = wc_all.abs().sum(dim=1).tolist()  # per branch: |A_raw[k] - bc[k]| <= sum_j |wc[k][j]| (one read-back per set of weights)
            ws, keep = [], [shared, wc_all, bc_all]
            for k in range(self.n_classes):
                own = dict(wc=wc_all[k], bc=bc_all[k:k + 1], wcls=Fn.f32c(self.classifiers[k].weight), bcls=Fn.f32c(self.classifiers[k].bias))
                w = N.ClamWeights()
                w.dtype, w.s0, w.s1, w.s2, w.n_classes = code, fc1.in_features, fc1.out_features, wa.out_features, 1
                for name, t in {**shared, **own}.items():
                    setattr(w, name, t.data_ptr())
                w.logit_bound = max(float(bounds[k]), 1e-30)
                nb = N.lib().hipt_clam_stream_packed_bytes(C_.byref(w))
                if nb:
                    own["stream_pk"] = torch.empty(nb, dtype=torch.uint8, device=device)
                    N.call("hipt_clam_stream_pack", C_.byref(w), N.ptr(own["stream_pk"]), N.stream_ptr(device))
                    w.stream_pk = own["stream_pk"].data_ptr()
                ws.append(w)
                keep.append(own)
            # all K branches in ONE pass over the bag (hipt_clam_mb_forward, round 6) where the streaming kernel takes the configuration:
            # the same shared tensors, wc / bc / the K one-row classifiers stacked, the image packed with its K rows of wc
            wm, mb = N.ClamWeights(), None
            wm.dtype, wm.s0, wm.s1, wm.s2 = code, fc1.in_features, fc1.out_features, wa.out_features
            wm.n_classes = wm.n_att = self.n_classes
            own = dict(wc=wc_all, bc=bc_all, wcls=Fn.f32c(torch.cat([c.weight for c in self.classifiers], dim=0)),
                       bcls=Fn.f32c(torch.cat([c.bias for c in self.classifiers], dim=0)))
            for name, t in {**shared, **own}.items():
                setattr(wm, name, t.data_ptr())
            wm.logit_bound = max(max(float(b) for b in bounds), 1e-30)
            nb = N.lib().hipt_clam_stream_packed_bytes(C_.byref(wm)) if 2 <= self.n_classes <= 4 else 0
            if nb:
                own["stream_pk"] = torch.empty(nb, dtype=torch.uint8, device=device)
                N.call("hipt_clam_stream_pack", C_.byref(wm), N.ptr(own["stream_pk"]), N.stream_ptr(device))
                wm.stream_pk = own["stream_pk"].data_ptr()
                if N.lib().hipt_clam_mb_supported(C_.byref(wm)):
                    mb = wm
            keep.append(own)
            self._packed = (key, ws, keep, mb)
        return self._packed[1]

    def _multi_infer(self, h, label, instance_eval, return_features, attention_only):
        """model_clam.py:226-264 without autograd: A [K, N] -> softmax over N per branch -> M [K, S1] -> logits[0, c] =
        classifiers[c](M[c]) -- branch by branch through the CLAM_SB inference kernels."""
        N.require_cuda(h, "CLAM_MB")
        if h.dim() != 2 or h.shape[0] == 0 or h.shape[1] != self._sizes[0]:
            raise ValueError(f"expected a non-empty [N, {self._sizes[0]}] bag, got {tuple(h.shape)}")
        ws = self._pack_branches(h.device)
        dev, K, n = h.device, self.n_classes, h.shape[0]
        bag = Fn.as_compute(h, ws[0].dtype)
        st = N.stream_ptr(dev)
        A_raw = torch.empty((K, n), dtype=torch.float32, device=dev)
        mb = self._packed[3] if self.one_pass else None
        S1 = self._sizes[1]
        if mb is not None:
            # ONE pass over the bag: gate once per row, K logits, h1 left in bf16 for the pooling kernel (two launches instead of K)
            scratch = Fn.workspace(dev, N.lib().hipt_clam_mb_workspace_bytes(C_.byref(mb), n), ("clam_mb", st.value), zero=True)
            if attention_only:
                N.call("hipt_clam_mb_forward", C_.byref(mb), N.ptr(bag), n, 1, N.ptr(A_raw), None, None, N.ptr(scratch), scratch.numel(), st)
                return A_raw
            M = torch.empty((K, S1), dtype=torch.float32, device=dev)
            logits = torch.empty((1, K), dtype=torch.float32, device=dev)
            N.call("hipt_clam_mb_forward", C_.byref(mb), N.ptr(bag), n, 0, N.ptr(A_raw), N.ptr(M), N.ptr(logits), N.ptr(scratch), scratch.numel(), st)
            return self._multi_finish(logits, A_raw, M, ws, bag, n, label, instance_eval, return_features)
        scratch = Fn.workspace(dev, N.lib().hipt_clam_workspace_bytes(C_.byref(ws[0]), n), ("clam", st.value), zero=True)
        if attention_only:
            for k, w in enumerate(ws):
                N.call("hipt_clam_sb_forward", C_.byref(w), N.ptr(bag), n, 1, N.ptr(A_raw[k]), None, None, None, None, N.ptr(scratch), scratch.numel(), st)
            return A_raw
        M = torch.empty((K, S1), dtype=torch.float32, device=dev)
        logits = torch.empty((1, K), dtype=torch.float32, device=dev)
        junk_p = torch.empty((K,), dtype=torch.float32, device=dev)      # (the one-class softmax / argmax of a branch: 1 and 0)
        junk_y = torch.empty((K,), dtype=torch.int64, device=dev)
        for k, w in enumerate(ws):
            N.call("hipt_clam_sb_forward", C_.byref(w), N.ptr(bag), n, 0, N.ptr(A_raw[k]), N.ptr(M[k]), N.ptr(logits[0, k:k + 1]), N.ptr(junk_p[k:k + 1]),
                   N.ptr(junk_y[k:k + 1]), N.ptr(scratch), scratch.numel(), st)
        return self._multi_finish(logits, A_raw, M, ws, bag, n, label, instance_eval, return_features)

    def _multi_finish(self, logits, A_raw, M, ws, bag, n, label, instance_eval, return_features):
        dev, K, S1, st = bag.device, self.n_classes, self._sizes[1], N.stream_ptr(bag.device)
        Y_hat = torch.topk(logits, 1, dim=1)[1]      # :251-252, on K numbers
        Y_prob = F.softmax(logits, dim=1)
        results = {}
        if instance_eval:
            # validate_clam (core_utils.py:506-560): top-k / bottom-k ids per branch on the device, only the selected h1 rows recomputed
            if self.k_sample > n:
                raise RuntimeError(f"selected index k out of range: k_sample={self.k_sample} > {n} rows (torch.topk, model_clam.py:120)")
            ids = torch.empty((K, 2, self.k_sample), dtype=torch.int64, device=dev)
            N.call("hipt_topk_rows", N.ptr(A_raw), K, n, self.k_sample, N.ptr(ids), st)
            rows = torch.empty((K, 2, self.k_sample, S1), dtype=torch.float32, device=dev)
            N.call("hipt_clam_gather_h1", C_.byref(ws[0]), N.ptr(bag), N.ptr(ids), 2 * K * self.k_sample, N.ptr(rows), st)
            results = self._instance_branch(lambda b: (rows[b, 0], rows[b, 1]), label)
        if return_features:
            results.update({'features': M})
        return logits, Y_prob, Y_hat, A_raw, results

    def _bag_logits(self, M):  # :248-250
        logits = torch.empty(1, self.n_classes).float().to(M.device)
        for c in range(self.n_classes):
            logits[0, c] = self.classifiers[c](M[c])
        return logits
