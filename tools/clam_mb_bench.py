"""CLAM_MB inference, 100 000 x 192 bf16, K = 3: HIP-event time per forward of the one-pass form (first pass alone = attention_only; both
passes) and of the branch-by-branch form.  python tools/clam_mb_bench.py [N=100000] [K=3] [S0=192]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hipt_abmil_atec23_amd import CLAM_MB, _native as N, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
s0 = int(sys.argv[3]) if len(sys.argv) > 3 else 192
dev = "cuda:0"
m = CLAM_MB(size_arg=[s0, 128, 64], k_sample=8, n_classes=K, subtyping=True)
m.load_state_dict(synth.make_state_dict(synth.clam_param_specs((s0, 128, 64), n_classes=K, multi=True), 193))
m.relocate()
m = m.eval().set_compute_dtype("bf16")
hb = synth.hash_uniform_torch((n, s0), 27, device=dev).bfloat16()


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


with torch.no_grad():
    m.one_pass = True
    print(f"N={n} K={K} S0={s0}: one pass, first pass alone (attention_only) {t(lambda: m(hb, attention_only=True)):.1f} us | both passes {t(lambda: m(hb)):.1f} us", flush=True)
    m.one_pass = False
    print(f"   branch by branch: attention_only {t(lambda: m(hb, attention_only=True)):.1f} us | full {t(lambda: m(hb)):.1f} us  (back-to-back launches, host overhead included)", flush=True)
