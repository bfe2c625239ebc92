"""The attention unit of one LayerNorm-chained block (hipt_vit_attention_unit) on random activation images: HIP-event time of the
fused QKV + attention kernel vs the QKV GEMM + attention kernel pair.  With the diagnostic library (make DEBUG_STAMPS=1,
HIPT_AMD_LIB=.../libhipt_abmil_dbg.so) HIPT_QKVATT_DBG=<mask> times the ablation variants of the fused kernel
(1 no weight DMA / ring syncs, 2 no fragment reads, 4 no attention phase, 8 no GEMM MFMAs: garbage results).
    python tools/attn_unit_bench.py [nseq=2048] [fused=1]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import _native as N, functional as Fn, synth
from hipt_abmil_atec23_amd.vision_transformer import vit_small

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
m = vit_small(patch_size=16, num_classes=0)
m.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
m = m.eval().to(dev).set_compute_dtype("bf16")
pk = m._tokens(synth.hash_uniform_torch((1, 3, 256, 256), 2, device=dev))[0]
M = nseq * 257
x = (torch.randn(M, 384, device=dev) * 1.0).bfloat16()
o = torch.empty_like(x)
ws = Fn.workspace(dev, N.lib().hipt_vit_workspace_bytes(pk.ref, nseq))
st = N.stream_ptr(dev)
run = lambda: N.call("hipt_vit_attention_unit", pk.ref, 3, N.ptr(x), nseq, N.ptr(o), fused, N.ptr(ws), ws.numel(), st)
for _ in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
a.record()
for _ in range(n):
    run()
b.record()
torch.cuda.synchronize()
us = a.elapsed_time(b) / n * 1e3
fl = 2 * M * 1152 * 384 + 4 * nseq * 6 * 257 * 257 * 64
print(f"attention unit nseq={nseq} fused={fused} dbg={os.environ.get('HIPT_QKVATT_DBG', '0')}: {us:.1f} us per call (incl. the [CLS]-row side launches), "
      f"{fl / us / 1e6:.0f} TFLOP/s, finite={bool(torch.isfinite(o.float()).all())}")
