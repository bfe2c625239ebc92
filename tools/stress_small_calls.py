"""Soak of the small-call path: calls of 1 .. 20 patches in both precisions, interleaved with a large call, repeated; every repeat must
return the bits of the first one (no race, no dependence on what ran before)."""
import sys

import torch

sys.path.insert(0, ".")
from hipt_abmil_atec23_amd import synth
from hipt_abmil_atec23_amd.vision_transformer import vit_small

m = vit_small().eval().cuda()
m.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
x = synth.hash_uniform_torch((48, 3, 256, 256), 5, device="cuda")
sizes = [1, 2, 3, 4, 5, 7, 16, 20, 48]
ref = {}
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    for dt in ("bf16", "fp32"):
        m.set_compute_dtype(dt)
        for n in sizes if dt == "bf16" else sizes[:6]:
            lo = (rep * 7 + n) % (48 - n + 1)
            out = m(x[lo:lo + n]).clone()
            key = (dt, n, lo)
            if key not in ref:
                ref[key] = out
            elif not torch.equal(ref[key], out):
                bad += 1
                print("MISMATCH", key, float((ref[key] - out).abs().max()), flush=True)
        # a patch alone and inside a small call: same bits (rows do not meet)
        a, b = m(x[5:6]), m(x[4:7])
        if not torch.equal(a, b[1:2]):
            bad += 1
            print("ROW DEPENDENCE", dt, float((a - b[1:2]).abs().max()), flush=True)
torch.cuda.synchronize()
print(f"{len(ref)} distinct calls, {bad} mismatches")
sys.exit(1 if bad else 0)
