"""The driver loop with its H2D hop in a FRESH process (bench.py starts this as a child and relays the JSON line as `extras.h2d_fresh_process`):
feature_store.extract_slide over one-region uint8 loader batches (8 gathered per call), batches resident in HBM vs in pinned host memory.
bench.py's own process has a dozen streams alive by the time it reaches its H2D leg, and HIP multiplexes streams over four hardware queues:
a copy queued behind a compute stream's kernels waits for them (0.88 of the resident loop there); a process that only runs the loop -- what an
extraction job is -- does not have that problem.      python tools/h2d_loop_bench.py [calls of 8 = 12] [coalesce = 8]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from hipt_abmil_atec23_amd import synth
from hipt_abmil_atec23_amd.feature_store import extract_slide

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 12
co = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
model, _, _ = B.build_models(dev, "bf16")
model.streams = 3
reg = synth.hash_uniform_torch((8, 3, 4096, 4096), 3, device=dev)
reg8 = ((reg * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
del reg
pinned = [reg8[i:i + 1].cpu().pin_memory() for i in range(8)]
out = {}
with tempfile.TemporaryDirectory() as td:
    for name, src in (("resident", [reg8[i:i + 1] for i in range(8)]), ("host", pinned)):
        ld = [(src[i], torch.tensor([[4096 * i, 0]])) for i in range(8)] * calls
        extract_slide(model, ld[:2 * co], td, "warm", coalesce=co)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        extract_slide(model, ld, td, "timed", coalesce=co)
        torch.cuda.synchronize()
        out[f"uint8_{name}_regions_per_s"] = len(ld) / (time.perf_counter() - t0)
out["coalesce"] = co
out["uint8_host_over_resident"] = out["uint8_host_regions_per_s"] / out["uint8_resident_regions_per_s"]
print("H2D_LOOP " + json.dumps(out), flush=True)
