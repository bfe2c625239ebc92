"""HIPT_4K.forward on R resident regions over S streams: regions/s (which S pays at which R).  python tools/streams_by_regions_bench.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from hipt_abmil_atec23_amd import synth

dev = torch.device("cuda:0")
model, _, _ = B.build_models(dev, "bf16")
reg = synth.hash_uniform_torch((24, 3, 4096, 4096), 3, device=dev)
for R in (2, 3, 4, 6, 8, 12, 16, 24):
    row = []
    for S in (1, 2, 3, 4):
        if S > R:
            continue
        model.streams = S
        x = reg[:R]
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        n = max(3, 48 // R)
        t0 = time.perf_counter()
        for _ in range(n):
            model(x)
        torch.cuda.synchronize()
        row.append(f"S={S}: {n * R / (time.perf_counter() - t0):6.1f}")
    print(f"R={R:2d}  " + "  ".join(row), flush=True)
