"""CLAM / ABMIL aggregator host mirror (reference: ``models/model_clam.py``).

``CLAM_SB`` and ``Attn_Net_Gated`` keep the reference's constructor arguments, attributes and
state-dict keys (``attention_net.0.*``, ``attention_net.{2|3}.attention_{a,b}.0.*``,
``...attention_c.*``, ``classifiers.*``, ``instance_classifiers.{c}.*``), so checkpoints written by
``utils/core_utils.py`` load with ``strict=True`` after the key cleaning of
``utils/eval_utils.py:51-57``.  Inference forwards (no autograd) run as ONE fused HIP pass over
the bag (csrc/abmil.hip).

Outside the accelerated path, by design ("next" row):
  * a forward that must be differentiated (``main.py`` trains this module) or that has active
    dropout runs as plain PyTorch ops ON THE SAME HIP DEVICE — this is the training path, not a
    fallback for inference; inference never takes it;
  * ``Attn_Net`` (ungated) and ``CLAM_MB`` are import-compatible PyTorch modules.
There is no CPU path for the HIP forwards: a CPU bag raises.
"""
from __future__ import annotations

import ctypes as C
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
        if self.attention_c.out_features != 1 or dropout_on or _needs_autograd(self, x):
            return self._torch_forward(x)  # training / multi-branch (CLAM_MB): PyTorch ops on the same device
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
            N.call("hipt_attn_net_gated", C.byref(w), N.ptr(xin), n, N.ptr(A), N.ptr(ws), ws.numel(), st)
        return A.reshape(*x.shape[:-1], 1), x


class CLAM_SB(nn.Module):
    """Single-branch CLAM / ABMIL (model_clam.py:77-191).

    ``size_arg`` is a key of the reference's size table (plus ``'hipt_384'``) or an explicit
    ``[S0, S1, S2]`` list."""

    def __init__(self, gate=True, size_arg="small", dropout=0.0, k_sample=8, n_classes=2,
                 instance_loss_fn=nn.CrossEntropyLoss(), subtyping=False):
        super().__init__()
        self.size_dict = dict(SIZE_DICT)
        size = list(size_arg) if isinstance(size_arg, (list, tuple)) else self.size_dict[size_arg]
        fc = [nn.Linear(size[0], size[1]), nn.ReLU()]
        if dropout > 0:
            fc.append(nn.Dropout(dropout))
        if gate:
            fc.append(Attn_Net_Gated(L=size[1], D=size[2], dropout=dropout, n_classes=1))
        else:
            fc.append(Attn_Net(L=size[1], D=size[2], dropout=dropout, n_classes=1))
        self.attention_net = nn.Sequential(*fc)
        self.classifiers = nn.Linear(size[1], n_classes)
        self.instance_classifiers = nn.ModuleList([nn.Linear(size[1], 2) for _ in range(n_classes)])
        self.k_sample = k_sample
        self.instance_loss_fn = instance_loss_fn
        self.n_classes = n_classes
        self.subtyping = subtyping
        self._gate = gate
        self._dropout = dropout
        self._compute_dtype = _default_dtype()
        self._packed = None
        initialize_weights(self)

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

    # ---- instance-level branches (model_clam.py:116-145); h_rows(ids) yields h1[ids] -------------
    def _inst_eval(self, A, h_rows, classifier):
        if A.dim() == 1:
            A = A.view(1, -1)
        top_p_ids = torch.topk(A, self.k_sample)[1][-1]
        top_n_ids = torch.topk(-A, self.k_sample, dim=1)[1][-1]
        device = A.device
        targets = torch.cat([self.create_positive_targets(self.k_sample, device),
                             self.create_negative_targets(self.k_sample, device)], dim=0)
        logits = classifier(h_rows(torch.cat([top_p_ids, top_n_ids], dim=0)))
        preds = torch.topk(logits, 1, dim=1)[1].squeeze(1)
        return self.instance_loss_fn(logits, targets), preds, targets

    def _inst_eval_out(self, A, h_rows, classifier):
        if A.dim() == 1:
            A = A.view(1, -1)
        top_p_ids = torch.topk(A, self.k_sample)[1][-1]
        targets = self.create_negative_targets(self.k_sample, A.device)
        logits = classifier(h_rows(top_p_ids))
        preds = torch.topk(logits, 1, dim=1)[1].squeeze(1)
        return self.instance_loss_fn(logits, targets), preds, targets

    def inst_eval(self, A, h, classifier):
        return self._inst_eval(A, lambda ids: torch.index_select(h, dim=0, index=ids), classifier)

    def inst_eval_out(self, A, h, classifier):
        return self._inst_eval_out(A, lambda ids: torch.index_select(h, dim=0, index=ids), classifier)

    def _instance_branch(self, A, h_rows, label):
        total, all_preds, all_targets = 0.0, [], []
        inst_labels = F.one_hot(label, num_classes=self.n_classes).squeeze()
        for i, classifier in enumerate(self.instance_classifiers):
            if inst_labels[i].item() == 1:
                loss, preds, targets = self._inst_eval(A, h_rows, classifier)
            elif self.subtyping:
                loss, preds, targets = self._inst_eval_out(A, h_rows, classifier)
            else:
                continue
            all_preds.extend(preds.cpu().numpy())
            all_targets.extend(targets.cpu().numpy())
            total += loss
        if self.subtyping:
            total /= len(self.instance_classifiers)
        return {'instance_loss': total, 'inst_labels': np.array(all_targets), 'inst_preds': np.array(all_preds)}

    # ---- training path: the reference op sequence as PyTorch ops on the module's device ----------
    def _torch_forward(self, h, label, instance_eval, return_features, attention_only):
        A, h = self.attention_net(h)
        A = torch.transpose(A, 1, 0)
        if attention_only:
            return A
        A_raw = A
        A = F.softmax(A, dim=1)
        results = self._instance_branch(A, lambda ids: torch.index_select(h, dim=0, index=ids), label) \
            if instance_eval else {}
        M = torch.mm(A, h)
        logits = self.classifiers(M)
        Y_hat = torch.topk(logits, 1, dim=1)[1]
        Y_prob = F.softmax(logits, dim=1)
        if return_features:
            results.update({'features': M})
        return logits, Y_prob, Y_hat, A_raw, results

    # ---- HIP path ---------------------------------------------------------------------------------
    def __getstate__(self):  # the packed weight image (ctypes struct + tensors) is a cache: never pickled / deep-copied
        d = self.__dict__.copy()
        d["_packed"] = None
        return d

    def _pack(self, device):
        code = N.dtype_code(self._compute_dtype)
        N.same_device("CLAM_SB", device, *self.parameters())  # e.g. relocate() never called: a clean error, not a GPU fault
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
            self._packed = (key, w, keep)
        return self._packed[1]

    def forward(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        dropout_on = self.training and self._dropout > 0
        if not self._gate or dropout_on or _needs_autograd(self, h):
            return self._torch_forward(h, label, instance_eval, return_features, attention_only)
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
        ws = Fn.workspace(dev, N.lib().hipt_clam_workspace_bytes(C.byref(w), n), ("clam", st.value), zero=True)
        if attention_only:
            N.call("hipt_clam_sb_forward", C.byref(w), N.ptr(bag), n, 1, N.ptr(A_raw), None, None, None, None,
                   N.ptr(ws), ws.numel(), st)
            return A_raw
        M = torch.empty((1, w.s1), dtype=torch.float32, device=dev)
        logits = torch.empty((1, w.n_classes), dtype=torch.float32, device=dev)
        Y_prob = torch.empty_like(logits)
        Y_hat = torch.empty((1, 1), dtype=torch.int64, device=dev)
        N.call("hipt_clam_sb_forward", C.byref(w), N.ptr(bag), n, 0, N.ptr(A_raw), N.ptr(M), N.ptr(logits), N.ptr(Y_prob),
               N.ptr(Y_hat), N.ptr(ws), ws.numel(), st)
        results = {}
        if instance_eval:
            # top-k stays on PyTorch-ROCm ops; only the 2k selected rows of h1 are
            # recomputed by the library instead of materialising h1 [N,S1]
            def h_rows(ids):
                ids = ids.to(torch.int64).contiguous()
                out = torch.empty((ids.numel(), w.s1), dtype=torch.float32, device=dev)
                N.call("hipt_clam_gather_h1", C.byref(w), N.ptr(bag), N.ptr(ids), ids.numel(), N.ptr(out), N.stream_ptr(dev))
                return out
            results = self._instance_branch(F.softmax(A_raw, dim=1), h_rows, label)
        if return_features:
            results.update({'features': M})
        return logits, Y_prob, Y_hat, A_raw, results


class CLAM_MB(CLAM_SB):
    """Multi-branch CLAM (model_clam.py:193-264): import-compatible PyTorch module, not on the HIP path."""

    def __init__(self, gate=True, size_arg="small", dropout=0.0, k_sample=8, n_classes=2,
                 instance_loss_fn=nn.CrossEntropyLoss(), subtyping=False):
        nn.Module.__init__(self)
        self.size_dict = dict(SIZE_DICT)
        size = list(size_arg) if isinstance(size_arg, (list, tuple)) else self.size_dict[size_arg]
        fc = [nn.Linear(size[0], size[1]), nn.ReLU()]
        if dropout > 0:
            fc.append(nn.Dropout(dropout))
        head = Attn_Net_Gated if gate else Attn_Net
        fc.append(head(L=size[1], D=size[2], dropout=dropout, n_classes=n_classes))
        self.attention_net = nn.Sequential(*fc)
        self.classifiers = nn.ModuleList([nn.Linear(size[1], 1) for _ in range(n_classes)])
        self.instance_classifiers = nn.ModuleList([nn.Linear(size[1], 2) for _ in range(n_classes)])
        self.k_sample = k_sample
        self.instance_loss_fn = instance_loss_fn
        self.n_classes = n_classes
        self.subtyping = subtyping
        initialize_weights(self)

    def forward(self, h, label=None, instance_eval=False, return_features=False, attention_only=False):
        device = h.device
        A, h = self.attention_net(h)
        A = torch.transpose(A, 1, 0)
        if attention_only:
            return A
        A_raw = A
        A = F.softmax(A, dim=1)
        results = {}
        if instance_eval:
            total, all_preds, all_targets = 0.0, [], []
            inst_labels = F.one_hot(label, num_classes=self.n_classes).squeeze()
            for i, classifier in enumerate(self.instance_classifiers):
                if inst_labels[i].item() == 1:
                    loss, preds, targets = self.inst_eval(A[i], h, classifier)
                elif self.subtyping:
                    loss, preds, targets = self.inst_eval_out(A[i], h, classifier)
                else:
                    continue
                all_preds.extend(preds.cpu().numpy())
                all_targets.extend(targets.cpu().numpy())
                total += loss
            if self.subtyping:
                total /= len(self.instance_classifiers)
            results = {'instance_loss': total, 'inst_labels': np.array(all_targets), 'inst_preds': np.array(all_preds)}
        M = torch.mm(A, h)
        logits = torch.empty(1, self.n_classes).float().to(device)
        for c in range(self.n_classes):
            logits[0, c] = self.classifiers[c](M[c])
        Y_hat = torch.topk(logits, 1, dim=1)[1]
        Y_prob = F.softmax(logits, dim=1)
        if return_features:
            results.update({'features': M})
        return logits, Y_prob, Y_hat, A_raw, results
