"""Which stage of HIPT_4K depends on how many regions share a call?  (bf16; 4096 x 4096 regions)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import HIPT_4K, synth
dev = "cuda:0"
m = HIPT_4K(None, None, dev, dev)
m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
m = m.eval().to(dev)
m.set_compute_dtype("bf16")
R = 8
x = torch.cat([synth.hash_uniform_torch((1, 3, 4096, 4096), 900 + i, device=dev) for i in range(R)])
for streams in (1, 2):
    m.streams = streams
    one = [m._run(x[i:i + 1], True) for i in range(R)]
    for n in (2, 3, 4, 5, 8):
        out, cls = m._run(x[:n], True)
        d_cls = [float((cls[i * 256:(i + 1) * 256] - one[i][1]).abs().max()) for i in range(n)]
        d_out = [float((out[i] - one[i][0][0]).abs().max()) for i in range(n)]
        print(f"streams={streams} n={n}: cls256 diff {['%.1e' % v for v in d_cls]}  out diff {['%.1e' % v for v in d_out]}")
# ViT-4K alone: same cls tokens, 1 vs n regions
tok = torch.stack([one[i][1] for i in range(R)])  # [R, 256, 384]
o1 = torch.cat([m.model4k.forward_tokens(tok[i:i + 1], 16, 16) for i in range(R)])
for n in (2, 4, 5, 8):
    on = m.model4k.forward_tokens(tok[:n].contiguous(), 16, 16)
    print(f"vit4k n={n}: diff {float((on - o1[:n]).abs().max()):.1e}")
