"""SHA-1 of HIPT_4K's bf16-mode features for a fixed 2-region input (and of ViT-256's [CLS] rows), per library build: to show that a kernel
change kept the bits.      HIPT_AMD_LIB=.../libhipt_abmil_old.so python tools/lib_output_hash.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from hipt_abmil_atec23_amd import synth

dev = torch.device("cuda:0")
model, clam, _ = B.build_models(dev, "bf16")
x = synth.hash_uniform_torch((2, 3, 4096, 4096), 11, device=dev)
with torch.no_grad():
    y = model(x)
    z = model(x[:1, :, :1024, :768])
h = lambda t: hashlib.sha1(t.float().cpu().numpy().tobytes()).hexdigest()[:16]
print(os.environ.get("HIPT_AMD_LIB", "shipped"), "features", h(y), "small region", h(z), "finite", bool(torch.isfinite(y).all()))
