"""One region per HIPT_4K.forward (the reference's batch size, extract_features_fp.py:159-171): time per call by the number of streams its
patches are spread over (HIPT_4K.patch_streams)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from hipt_abmil_atec23_amd import HIPT_4K, synth

dev = "cuda:0"
m = HIPT_4K(None, None, dev, dev)
m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
m = m.eval().to(dev)
m.set_compute_dtype("bf16")
x = synth.hash_uniform_torch((1, 3, 4096, 4096), 3, device=dev)


def timed(n=20):
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        m(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for rep in range(2):
    for streams in [int(a) for a in sys.argv[1:] if a.isdigit()] or [2, 1]:
        m.patch_streams = streams  # (one region: its patches over that many streams; the class default is 1)
        a = timed()
        print(f"streams {streams}: one region per call {a:.3f} ms ({1e3 / a:.1f} regions/s)", flush=True)
