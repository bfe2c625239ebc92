"""extract_slide's gathering loop over resident one-region batches: regions/s by HIPT_4K.streams and coalesce.  python tools/gather_loop_bench.py"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from hipt_abmil_atec23_amd import synth
from hipt_abmil_atec23_amd.feature_store import extract_slide

dev = torch.device("cuda:0")
model, _, _ = B.build_models(dev, "bf16")
reg = synth.hash_uniform_torch((8, 3, 4096, 4096), 3, device=dev)
with tempfile.TemporaryDirectory() as td:
    for co in (8, 16, 24):
        for streams in (1, 2, 3, 4):
            model.streams = streams
            ld = [(reg[i % 8:i % 8 + 1], torch.tensor([[4096 * i, 0]])) for i in range(4 * co)]
            extract_slide(model, ld[:co], td, "warm", coalesce=co)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            extract_slide(model, ld, td, "timed", coalesce=co)
            torch.cuda.synchronize()
            print(f"coalesce {co:2d} streams {streams}: {len(ld) / (time.perf_counter() - t0):.1f} regions/s", flush=True)
