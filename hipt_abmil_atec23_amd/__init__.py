"""hipt_abmil_atec23_amd — MI355X-native (gfx950) implementation of the one data-parallel hot path of
scjjb/HIPT_ABMIL_ATEC23: HIPT_4K feature extraction (ViT-256 -> ViT-4K) and the CLAM_SB / ABMIL
gated-attention pooling, behind the reference's own Python call surface.

    from hipt_abmil_atec23_amd import HIPT_4K, CLAM_SB, Attn_Net_Gated
    import hipt_abmil_atec23_amd as amd; amd.install()   # make the reference scripts import these classes

All compute runs in hand-written HIP kernels loaded from ``libhipt_abmil.so`` (C ABI:
``include/hipt_abmil.h``); there is no CPU or eager fallback for inference.
"""
from . import synth  # noqa: F401  (no torch import at package import time)

__all__ = ["HIPT_4K", "CLAM_SB", "CLAM_MB", "Attn_Net", "Attn_Net_Gated", "VisionTransformer",
           "VisionTransformer4K", "vit_small", "vit4k_xs", "install", "build_native"]

_LAZY = {
    "HIPT_4K": ("hipt_4k", "HIPT_4K"),
    "CLAM_SB": ("model_clam", "CLAM_SB"), "CLAM_MB": ("model_clam", "CLAM_MB"),
    "Attn_Net": ("model_clam", "Attn_Net"), "Attn_Net_Gated": ("model_clam", "Attn_Net_Gated"),
    "VisionTransformer": ("vision_transformer", "VisionTransformer"), "vit_small": ("vision_transformer", "vit_small"),
    "VisionTransformer4K": ("vision_transformer4k", "VisionTransformer4K"), "vit4k_xs": ("vision_transformer4k", "vit4k_xs"),
    "install": ("dropin", "install"), "build_native": ("_native", "build"),
    "FeatureWriter": ("feature_store", "FeatureWriter"), "extract_slide": ("feature_store", "extract_slide"),
    "load_bag": ("feature_store", "load_bag"), "load_coords": ("feature_store", "load_coords"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(f"{__name__}.{mod}"), attr)
    raise AttributeError(name)
