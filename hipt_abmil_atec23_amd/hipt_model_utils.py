"""Loaders / transforms of the HIPT_4K extractor (reference: ``HIPT_4K/hipt_model_utils.py``).

The reference file as committed cannot even be imported (TabError at :72 and :109); this follows
its intended behaviour: build the architecture with the factory defaults (``img_size=[224]`` ->
197-row pos_embed), freeze, ``eval()``, load the DINO ``teacher`` dict after stripping
``module.`` / ``backbone.`` prefixes with ``strict=False`` (:39-73, :76-110).

Extension (there are no checkpoints offline): ``pretrained_weights=None`` builds the model with
its random initialisation instead of asserting.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import vision_transformer as vits
from . import vision_transformer4k as vits4k


def _load_dino(model, pretrained_weights, checkpoint_key="teacher"):
    state_dict = torch.load(pretrained_weights, map_location="cpu")
    if checkpoint_key is not None and checkpoint_key in state_dict:
        print(f"Take key {checkpoint_key} in provided checkpoint dict")
        state_dict = state_dict[checkpoint_key]
    state_dict = {k.replace("module.", "").replace("backbone.", ""): v for k, v in state_dict.items()}
    msg = model.load_state_dict(state_dict, strict=False)
    print('Pretrained weights found at {} and loaded with msg: {}'.format(pretrained_weights, msg))
    return model


def _build(factory, pretrained_weights, **kw):
    model = factory(**kw)
    for p in model.parameters():
        p.requires_grad = False
    model.eval()
    if pretrained_weights is None:
        return model
    assert os.path.isfile(pretrained_weights), "pretrained weights not available at {}".format(pretrained_weights)
    return _load_dino(model, pretrained_weights)


def get_vit256(pretrained_weights, arch='vit_small', device=torch.device('cuda:0')):
    """ViT-256 (hipt_model_utils.py:39-73); returned on CPU like the reference, caller moves it."""
    return _build(vits.__dict__[arch], pretrained_weights, patch_size=16, num_classes=0)


def get_vit4k(pretrained_weights, arch='vit4k_xs', device=torch.device('cuda:1')):
    """ViT-4K (hipt_model_utils.py:76-110)."""
    return _build(vits4k.__dict__[arch], pretrained_weights, num_classes=0)


class _EvalTransform:
    """ToTensor + Normalize(mean=0.5, std=0.5) (hipt_model_utils.py:113-118) without torchvision:
    HxWxC uint8 (PIL image or ndarray) -> CxHxW float32 in [-1, 1]."""

    def __call__(self, img):
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)
        t = t.float().div(255.0) if t.dtype == torch.uint8 else t.float()
        return (t - 0.5) / 0.5


def eval_transforms():
    return _EvalTransform()


def tensorbatch2im(input_image, imtype=np.uint8):
    """(B,C,W,H) tensor in [-1,1] -> (B,W,H,C) uint8 array (hipt_model_utils.py:137-154)."""
    if isinstance(input_image, np.ndarray):
        return input_image.astype(imtype)
    a = input_image.cpu().float().numpy()
    return ((np.transpose(a, (0, 2, 3, 1)) + 1) / 2.0 * 255.0).astype(imtype)
