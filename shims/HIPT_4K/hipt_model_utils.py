# Overlay for a reference checkout: replaces HIPT_4K/hipt_model_utils.py with the gfx950 implementation.
from hipt_abmil_atec23_amd.hipt_model_utils import *  # noqa: F401,F403
from hipt_abmil_atec23_amd import hipt_model_utils as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
