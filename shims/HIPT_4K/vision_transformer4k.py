# Overlay for a reference checkout: replaces HIPT_4K/vision_transformer4k.py with the gfx950 implementation.
from hipt_abmil_atec23_amd.vision_transformer4k import *  # noqa: F401,F403
from hipt_abmil_atec23_amd import vision_transformer4k as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
