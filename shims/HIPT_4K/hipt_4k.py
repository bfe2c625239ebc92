# Overlay for a reference checkout: replaces HIPT_4K/hipt_4k.py with the gfx950 implementation.
from hipt_abmil_atec23_amd.hipt_4k import *  # noqa: F401,F403
from hipt_abmil_atec23_amd import hipt_4k as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
