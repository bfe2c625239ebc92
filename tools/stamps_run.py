"""In-kernel stamp report of the pipelined kernels inside the real pipeline (8 regions, bf16):
    HIPT_SEQGEMM_STAMPS=1 python tools/stamps_run.py 2>&1 | grep -A200 "second call"
The stamps describe the FIRST tile of every workgroup (all CUs in step: row phases at their most congested)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import HIPT_4K, synth
dev = "cuda:0"
m = HIPT_4K(None, None, dev, dev)
m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
m = m.eval().to(dev).set_compute_dtype("bf16")
m.streams = int(os.environ.get("STREAMS", m.streams))  # (STREAMS=1: all 8 regions = 2 048 patches in every launch)
x = synth.hash_uniform_torch((8, 3, 4096, 4096), 3, device=dev)
with torch.no_grad():
    m(x); torch.cuda.synchronize()
    print("=== second call", file=sys.stderr)
    m(x); torch.cuda.synchronize()
