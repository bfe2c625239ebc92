"""HIPT_4K.forward on ONE 4096x4096 region per call (the reference's batch size) for several stream counts."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import HIPT_4K, synth
dev = torch.device("cuda:0")
m = HIPT_4K(None, None, dev, dev)
m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
m = m.eval().to(dev).set_compute_dtype("bf16")
x = synth.hash_uniform_torch((1, 3, 4096, 4096), 3, device=dev)
ref = None
for streams in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    m.streams = streams
    for _ in range(3):
        out = m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref = out.clone()
    print(f"streams={streams}: {1 / dt:.1f} regions/s ({dt * 1e3:.2f} ms per region); equal to streams=1 bit for bit: {bool(torch.equal(out, ref))}")
