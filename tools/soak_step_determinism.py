"""Run-to-run determinism of the bench step under stream concurrency (round 6: the uint8 embedding race showed only when two streams ran at once):
HIPT_4K.forward on R resident regions over S streams, repeated; every repeat must return the first one's bits.  fp32 and uint8 (interleaved) input,
CLAM_SB on a 100 000 x 384 bag beside it.      python tools/soak_step_determinism.py [R=24] [reps=20]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from hipt_abmil_atec23_amd import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
model, clam, _ = B.build_models(dev, "bf16")
region = synth.hash_uniform_torch((R, 3, 4096, 4096), 3, device=dev)
reg8 = ((region * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
bag = synth.hash_uniform_torch((100000, 384), 40, device=dev).to(torch.bfloat16)
bad = 0
for streams in (3, 2, 1):
    model.streams = streams
    for name, x in (("fp32", region), ("uint8", reg8)):
        ref = model(x).clone()
        with torch.no_grad():
            lref = [t.clone() for t in clam(bag)[:4]]
        n_bad = 0
        for i in range(reps):
            o = model(x)
            with torch.no_grad():
                l = clam(bag)[:4]
            torch.cuda.synchronize()
            if not torch.equal(o, ref) or not all(torch.equal(a, b) for a, b in zip(l, lref)):
                n_bad += 1
                if n_bad <= 2:
                    rows = [r for r in range(R) if not torch.equal(o[r], ref[r])]
                    print(f"   MISMATCH streams {streams} {name} rep {i}: regions {rows}", flush=True)
        bad += n_bad
        print(f"streams {streams} {name}: {n_bad} / {reps} repeats differ", flush=True)
print("TOTAL mismatches:", bad)
