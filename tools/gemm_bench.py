"""Micro-benchmark of hipt_linear / hipt_attention / layernorm on the ViT-256 shapes (M = 256 x 257)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hipt_abmil_atec23_amd import _native as N, functional as Fn
import ctypes as C

dev = torch.device("cuda:0")
P, NT, D, H = 256, 257, 384, 1536
M = P * NT
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = sys.argv[2] if len(sys.argv) > 2 else ""
dt = torch.bfloat16
code = N.HIPT_BF16

def lin(name, K, Nn, flags, resid=False, out_f32=False):
    if only and only not in name:
        return
    a = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(Nn, K, device=dev) * 0.05).to(dt)
    b = torch.randn(Nn, device=dev)
    r = torch.randn(M, Nn, device=dev) if resid else None
    out = torch.empty(M, Nn, device=dev, dtype=torch.float32 if out_f32 else dt)
    lib = N.lib()
    def run():
        N.check(lib.hipt_linear(N.ptr(a), K, N.ptr(w), K, N.ptr(b), N.ptr(r), N.ptr(out), Nn, M, Nn, K, code, flags, N.stream_ptr(dev)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"{name:10s} M={M} N={Nn:5d} K={K:5d}: {us:8.1f} us  {2*M*Nn*K/us/1e6:8.1f} TFLOP/s")

lin("qkv", D, 3 * D, 0)
lin("proj", D, D, N.EPI_RESID | N.EPI_OUT_F32, resid=True, out_f32=True)
lin("fc1", D, H, N.EPI_GELU)
lin("fc2", H, D, N.EPI_RESID | N.EPI_OUT_F32, resid=True, out_f32=True)
lin("fc1plain", D, H, 0)
