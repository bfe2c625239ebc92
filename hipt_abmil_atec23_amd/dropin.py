"""Drop-in hook: make the reference's scripts import this package's classes.

The reference imports the hot-path classes by module path:
  ``from HIPT_4K.hipt_4k import HIPT_4K``, ``from HIPT_4K.hipt_model_utils import eval_transforms``
  (extract_features_fp.py:15-16, create_heatmaps.py:23), ``import HIPT_4K.vision_transformer as vits``
  (hipt_4k.py:25-26) and ``from models.model_clam import CLAM_MB, CLAM_SB`` (utils/core_utils.py:7,
  utils/eval_utils.py:6).  ``install()`` registers this package's modules in ``sys.modules`` under
  exactly those names BEFORE the reference scripts import them; the reference's ``models`` package
  (which also holds model_mil.py / resnet_custom.py) is left alone — only the ``model_clam``
  sub-module is replaced.  The alternative is the overlay files under ``shims/``.
"""
from __future__ import annotations

import importlib
import importlib.util
import sys
import types

_MAP = {
    "HIPT_4K.hipt_4k": "hipt_4k",
    "HIPT_4K.hipt_model_utils": "hipt_model_utils",
    "HIPT_4K.vision_transformer": "vision_transformer",
    "HIPT_4K.vision_transformer4k": "vision_transformer4k",
    "models.model_clam": "model_clam",
}


def install(verbose: bool = False):
    """Register the HIP-backed modules under the reference's import paths. Returns the mapping."""
    done = {}
    pkg = __name__.rsplit(".", 1)[0]
    for ref_name, ours in _MAP.items():
        mod = importlib.import_module(f"{pkg}.{ours}")
        parent_name = ref_name.split(".")[0]
        if parent_name not in sys.modules:
            # HIPT_4K/: a directory of four files we replace entirely -> always a stub namespace.
            # models/: the reference's real package (model_mil.py, resnet_custom.py ...) must keep working,
            # so it is imported if it is on sys.path and only stubbed when there is no reference checkout.
            spec = None
            if parent_name != "HIPT_4K":
                try:
                    spec = importlib.util.find_spec(parent_name)
                except (ImportError, ValueError):
                    spec = None
            if spec is not None:
                importlib.import_module(parent_name)
            else:
                parent = types.ModuleType(parent_name)
                parent.__path__ = []
                parent.__hipt_amd_stub__ = True
                sys.modules[parent_name] = parent
        sys.modules[ref_name] = mod
        parent = sys.modules.get(parent_name)
        if parent is not None:
            setattr(parent, ref_name.split(".")[1], mod)
        done[ref_name] = mod.__name__
        if verbose:
            print(f"[hipt_abmil_atec23_amd] {ref_name} -> {mod.__name__}")
    return done


def uninstall():
    for ref_name in _MAP:
        sys.modules.pop(ref_name, None)
    for parent in ("HIPT_4K", "models"):
        if getattr(sys.modules.get(parent), "__hipt_amd_stub__", False):
            sys.modules.pop(parent, None)
