# Overlay for a reference checkout: replaces models/model_clam.py with the gfx950 implementation.
from hipt_abmil_atec23_amd.model_clam import *  # noqa: F401,F403
from hipt_abmil_atec23_amd import model_clam as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
