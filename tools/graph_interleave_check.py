import sys, torch
sys.path.insert(0, ".")
from hipt_abmil_atec23_amd import HIPT_4K, synth
dev = "cuda:0"
m = HIPT_4K(None, None, dev, dev)
m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
m = m.eval().to(dev); m.set_compute_dtype("bf16")
x = synth.hash_uniform_torch((1, 3, 4096, 4096), 3, device=dev)
p = synth.hash_uniform_torch((1, 3, 256, 256), 5, device=dev)
step = int(sys.argv[1])
m.model4k.graph_max_regions = 2 if step >= 2 else 0
ref = m.model256(p); torch.cuda.synchronize(); print("small captured", flush=True)
for streams in (2, 1, 3, 2):
    m.streams = streams
    for _ in range(3): o = m(x)
    torch.cuda.synchronize(); print("big ok, streams", streams, flush=True)
    assert torch.equal(m.model256(p), ref); torch.cuda.synchronize(); print("small replay ok", flush=True)
print("done")
