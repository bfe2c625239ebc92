"""Does an H2D copy from pinned memory overlap the ViT kernels on this stack?  (round 6: the copy engine question behind feature_store._HostFeed)
    python tools/h2d_overlap_probe.py
Prints: the copy alone (GB/s), the compute alone (ms per 8 uint8 regions), both issued together on two streams (ms), and the same with
HSA_ENABLE_SDMA as found in the environment."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from hipt_abmil_atec23_amd import synth

dev = torch.device("cuda:0")
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"), flush=True)
model, _, _ = B.build_models(dev, "bf16")
model.streams = 3
reg = synth.hash_uniform_torch((8, 3, 4096, 4096), 3, device=dev)
reg8 = ((reg * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
del reg
host = reg8.cpu().pin_memory()
dst = torch.empty_like(reg8)
cs = torch.cuda.Stream()


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def copy():
    with torch.cuda.stream(cs):
        dst.copy_(host, non_blocking=True)


def compute():
    model(reg8)


def both():
    copy()
    compute()


c, k, b = t(copy), t(compute), t(both)
print(f"copy alone {c:.2f} ms ({host.numel() / c / 1e6:.1f} GB/s) | compute alone {k:.2f} ms | both {b:.2f} ms (sum {c + k:.2f}: overlap hides {c + k - b:.2f} ms of the copy)", flush=True)
# one region at a time (the loader's batches): eight copies of 50 MB
hs = [host[i:i + 1].clone().pin_memory() for i in range(8)]


def copy8():
    with torch.cuda.stream(cs):
        for i in range(8):
            dst[i:i + 1].copy_(hs[i], non_blocking=True)


def both8():
    copy8()
    compute()


c8, b8 = t(copy8), t(both8)
print(f"eight 50 MB copies {c8:.2f} ms | with compute {b8:.2f} ms (sum {c8 + k:.2f})", flush=True)

# ---- the loop's own shape: double-buffered gather buffers, events both ways, read-back one call late ----
bufs = [torch.empty_like(reg8), torch.empty_like(reg8)]
rs = torch.cuda.Stream()


def loop(n=12, wait_free=True, readback=True, late=True):
    free = [None, None]
    pend = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        s = k & 1
        if wait_free and free[s] is not None:
            cs.wait_event(free[s])
        with torch.cuda.stream(cs):
            for i in range(8):
                bufs[s][i:i + 1].copy_(hs[i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(cs)
        torch.cuda.current_stream().wait_event(ev)
        out = model(bufs[s])
        f = torch.cuda.Event()
        f.record()
        free[s] = f
        if readback:
            pend.append((out, f))
            while len(pend) > (1 if late else 0):
                o, e = pend.pop(0)
                rs.wait_event(e)
                with torch.cuda.stream(rs):
                    h = o.to("cpu", non_blocking=True)
                rs.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for kw in (dict(wait_free=False, readback=False), dict(wait_free=True, readback=False), dict(wait_free=True, readback=True, late=True),
           dict(wait_free=True, readback=True, late=False)):
    loop(4, **kw)
    print(kw, f"{loop(12, **kw):.2f} ms per call of 8 regions", flush=True)
import tempfile
from hipt_abmil_atec23_amd.feature_store import extract_slide
ld = [(hs[i], torch.tensor([[4096 * i, 0]])) for i in range(8)] * 12
with tempfile.TemporaryDirectory() as td:
    extract_slide(model, ld[:16], td, "warm")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    extract_slide(model, ld, td, "timed")
    torch.cuda.synchronize()
    print(f"extract_slide, host batches: {(time.perf_counter() - t0) / 12 * 1e3:.2f} ms per call of 8 regions", flush=True)
