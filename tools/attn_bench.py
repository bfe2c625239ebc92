"""Micro-benchmark / cross-check of hipt_attention on the ViT-256 shape (B x 257 tokens, 6 heads x 64, bf16).
HIPT_ATTN_V1=1 selects the generic kernel; run both and compare the saved outputs:
    python tools/attn_bench.py 2048 /tmp/a2.pt;  HIPT_ATTN_V1=1 python tools/attn_bench.py 2048 /tmp/a1.pt /tmp/a2.pt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import _native as N, functional as Fn, synth

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ntok, heads, dh = 257, 6, 64
qkv = synth.hash_uniform_torch((B, ntok, 3 * heads * dh), 51, 1.5, device=dev).to(torch.bfloat16)
out, _ = Fn.attention(qkv, heads, dh ** -0.5, dtype=N.HIPT_BF16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 10
e0.record()
for _ in range(iters):
    Fn.attention(qkv, heads, dh ** -0.5, dtype=N.HIPT_BF16)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
flops = 4.0 * B * heads * ntok * ntok * dh
print(f"{'v1' if os.environ.get('HIPT_ATTN_V1') else 'v2'} B={B}: {us:.1f} us  {flops / us / 1e6:.1f} TFLOP/s  finite={bool(torch.isfinite(out.float()).all())}")
if len(sys.argv) > 2:
    torch.save(out.cpu(), sys.argv[2])
if len(sys.argv) > 3:
    ref = torch.load(sys.argv[3]).float()
    d = (out.cpu().float() - ref)
    print("max |diff| vs", sys.argv[3], float(d.abs().max()), "rel-L2", float(d.norm() / ref.norm()))
