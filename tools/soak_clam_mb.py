"""CLAM_MB one-pass inference against the branch-by-branch form over random bag lengths, branch counts and widths (round 6 soak).
python tools/soak_clam_mb.py [cases=60]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hipt_abmil_atec23_amd import CLAM_MB, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = "cuda:0"
rng = random.Random(7)
models = {}
worst = [0.0, 0.0, 0.0]
bad = 0
for c in range(cases):
    K, s0 = rng.choice((2, 3, 4)), rng.choice((192, 384))
    n = rng.choice((rng.randint(1, 70), rng.randint(71, 5000), rng.randint(5001, 150000)))
    if (K, s0) not in models:
        m = CLAM_MB(size_arg=[s0, 128, 64], k_sample=1, n_classes=K, subtyping=True)
        m.load_state_dict(synth.make_state_dict(synth.clam_param_specs((s0, 128, 64), n_classes=K, multi=True), 190 + K))
        m.relocate()
        models[(K, s0)] = m.eval().set_compute_dtype("bf16")
    m = models[(K, s0)]
    h = synth.hash_uniform_torch((n, s0), 500 + c, device=dev)
    with torch.no_grad():
        m.one_pass = True
        lg, yp, yh, a, r = m(h, return_features=True)
        m.one_pass = False
        lg2, yp2, yh2, a2, r2 = m(h, return_features=True)
        m.one_pass = True
    da = float((a - a2).abs().max())
    dm = float((r["features"] - r2["features"]).norm() / r2["features"].norm())
    dl = float((lg - lg2).abs().max())
    worst = [max(worst[0], da), max(worst[1], dm), max(worst[2], dl)]
    ok = da < 1e-4 and dm < 2e-3 and dl < 2e-3 and bool(torch.isfinite(lg).all())
    bad += 0 if ok else 1
    if not ok:
        print(f"MISMATCH K={K} S0={s0} N={n}: A_raw {da:.2e} M {dm:.2e} logits {dl:.2e}", flush=True)
print(f"{cases} cases, {bad} outside the bars; worst A_raw {worst[0]:.2e}, M rel-L2 {worst[1]:.2e}, logits {worst[2]:.2e}")
