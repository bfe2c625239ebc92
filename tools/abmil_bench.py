"""Micro-benchmark of CLAM_SB.forward on a 100000 x 384 bf16 bag (rotating bags > Infinity Cache)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import CLAM_SB, synth
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
c = CLAM_SB(size_arg="hipt_384")
c.load_state_dict(synth.make_state_dict(synth.clam_param_specs((384, 128, 64)), 384))
c = c.eval().to(dev).set_compute_dtype("bf16")
bags = [synth.hash_uniform_torch((n, 384), 40 + i, device=dev).bfloat16() for i in range(5)]
with torch.no_grad():
    for i in range(5): c(bags[i % 5])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): c(bags[i % 5])
    e1.record(); torch.cuda.synchronize()
print(f"CLAM_SB bf16 N={n}: {e0.elapsed_time(e1)/20*1e3:.1f} us per forward (fused + combine + launch gaps)")
