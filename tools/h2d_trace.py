"""Where a gathered call's time goes in extract_slide's host-fed loop: HIP events on the copy stream (first copy issued -> last copy done) and on the
compute stream (call start -> call end), per call, against one reference event.  python tools/h2d_trace.py"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B
from hipt_abmil_atec23_amd import feature_store as FS, synth

dev = torch.device("cuda:0")
model, _, _ = B.build_models(dev, "bf16")
model.streams = 3
reg = synth.hash_uniform_torch((8, 3, 4096, 4096), 3, device=dev)
reg8 = ((reg * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
del reg
pinned = [reg8[i:i + 1].cpu().pin_memory() for i in range(8)]
rec = []
E = lambda: torch.cuda.Event(enable_timing=True)
t_ref = E()
orig_begin, orig_finish = FS._HostFeed.begin, FS._HostFeed.finish


def begin(self, cap, like):
    orig_begin(self, cap, like)
    e = E()
    e.record(self.copy_stream)
    self._t_copy0 = e


def finish(self):
    e = E()
    e.record(self.copy_stream)
    out = orig_finish(self)
    c0 = E()
    c0.record(torch.cuda.current_stream(self.device))
    rec.append({"copy0": self._t_copy0, "copy1": e, "comp0": c0, "host_finish": time.perf_counter()})
    return out


orig_release = FS._HostFeed.release


def release(self, s):
    orig_release(self, s)
    e = E()
    e.record(torch.cuda.current_stream(self.device))
    rec[-1]["comp1"] = e
    rec[-1]["host_release"] = time.perf_counter()


FS._HostFeed.begin, FS._HostFeed.finish, FS._HostFeed.release = begin, finish, release
ld = [(pinned[i], torch.tensor([[4096 * i, 0]])) for i in range(8)] * 8
with tempfile.TemporaryDirectory() as td:
    FS.extract_slide(model, ld[:16], td, "warm")
    torch.cuda.synchronize()
    rec.clear()
    t_ref.record()
    h0 = time.perf_counter()
    FS.extract_slide(model, ld, td, "timed")
    torch.cuda.synchronize()
    print(f"total {(time.perf_counter() - h0) * 1e3:.1f} ms for {len(ld)} regions")
for k, r in enumerate(rec):
    f = lambda e: t_ref.elapsed_time(e)
    print(f"call {k}: copies {f(r['copy0']):7.1f} -> {f(r['copy1']):7.1f} ms | compute {f(r['comp0']):7.1f} -> {f(r['comp1']):7.1f} ms ({f(r['comp1']) - f(r['comp0']):.1f}) | host: finish at {(r['host_finish'] - h0) * 1e3:7.1f}, model() returned at {(r['host_release'] - h0) * 1e3:7.1f}")
