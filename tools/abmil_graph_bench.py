"""CLAM_SB.forward on the 100000 x 384 bf16 bag, K forwards captured in one HIP graph and replayed between ONE event pair:
the kernel's own pace (no host time between launches, no per-launch event records).  Rotating bags > Infinity Cache."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import CLAM_SB, synth
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = 20
c = CLAM_SB(size_arg="hipt_384")
c.load_state_dict(synth.make_state_dict(synth.clam_param_specs((384, 128, 64)), 384))
c = c.eval().to(dev).set_compute_dtype("bf16")
bags = [synth.hash_uniform_torch((n, 384), 40 + i, device=dev).bfloat16() for i in range(5)]
with torch.no_grad():
    for i in range(5): c(bags[i % 5])
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3): c(bags[i % 5])
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        outs = [c(bags[i % 5]) for i in range(K)]
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / K * 1e3)
    ref = c(bags[0])
    print(f"CLAM_SB bf16 N={n}: {best:.1f} us per forward in a replayed graph of {K} (min of 5); logits equal to a plain call: "
          f"{torch.equal(outs[0][0], ref[0])}")
