import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import HIPT_4K, synth
from hipt_abmil_atec23_amd import _native as N
dev = torch.device("cuda:0")
m = HIPT_4K(None, None, dev, dev)
m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
m = m.eval().to(dev).set_compute_dtype("bf16")
x = synth.hash_uniform_torch((1, 3, 4096, 4096), 3, device=dev)
for streams in (1, 3):
    m.streams = streams
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize()
    print(f"streams={streams}: {(time.perf_counter()-t0)/10*1e3:.2f} ms per region")
m.streams = 1
N.profile_enable(True)
for _ in range(5): m(x)
torch.cuda.synchronize()
prof = N.profile_read()
N.profile_enable(False)
tot = 0
for cat, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print(f"{cat:24s} {cnt/5:6.1f} launches/region  {ms/cnt*1e3:9.1f} us avg  {ms/5:8.3f} ms/region")
    tot += ms / 5
print("sum", tot)
