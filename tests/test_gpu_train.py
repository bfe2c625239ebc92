"""GPU tests (-m gpu) of the CLAM training step on the HIP kernels (csrc/clam_train.hip): forward
outputs and the gradient of EVERY parameter against the reference's own autograd (tests/golden/clam_*grad*.npz, made by
importing the reference's CLAM_SB / CLAM_MB) -- fp32, 1e-4 -- and, for sizes / dropout masks the fixtures do not hold,
against the PyTorch-CPU oracle's autograd step on the same inputs."""
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden
from hipt_abmil_atec23_amd import _native as N
from hipt_abmil_atec23_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4

TRAIN_CASES = [
    # name, size, base, (N, S0), seed, label, n_classes, multi, k, subtyping, instance_eval
    ("clam_grad_hipt_big_n15", (192, 128, 64), 192, (15, 192), 21, 1, 2, False, 8, False, True),
    ("clam_grad_hipt_big_n100", (192, 128, 64), 192, (100, 192), 22, 1, 2, False, 8, False, True),
    ("clam_grad_hipt_big_n2000", (192, 128, 64), 192, (2000, 192), 23, 1, 2, False, 8, False, True),
    ("clam_grad_hipt_big_n100_bagonly", (192, 128, 64), 192, (100, 192), 22, 0, 2, False, 8, False, False),
    ("clam_grad_hipt_smallest_n100", (192, 8, 4), 8, (100, 192), 6, 0, 2, False, 4, True, True),
    ("clam_mb_grad_hipt_big_n100", (192, 128, 64), 193, (100, 192), 24, 2, 3, True, 8, True, True),
]


def md(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0


def make(size, base, ncls, multi, k, subtyping, dropout=0.0):
    from hipt_abmil_atec23_amd import CLAM_MB, CLAM_SB
    m = (CLAM_MB if multi else CLAM_SB)(size_arg=list(size), dropout=dropout, k_sample=k, n_classes=ncls, subtyping=subtyping)
    m.load_state_dict(synth.make_state_dict(synth.clam_param_specs(size, n_classes=ncls, multi=multi, dropout=dropout > 0), base))
    m.relocate()
    return m.train()


def step(m, h, label, inst, bag_weight=0.7):
    m.zero_grad()
    lab = torch.tensor([label], device=h.device)
    logits, y_prob, y_hat, a_raw, res = m(h, label=lab, instance_eval=inst, return_features=True)
    loss = F.cross_entropy(logits, lab)
    total = bag_weight * loss + (1 - bag_weight) * res["instance_loss"] if inst else loss
    total.backward()
    return logits, y_prob, y_hat, a_raw, res, total


@pytest.mark.parametrize("name,size,base,shape,seed,label,ncls,multi,k,subtyping,inst", TRAIN_CASES)
def test_train_step_vs_reference_gradients(name, size, base, shape, seed, label, ncls, multi, k, subtyping, inst):
    g = golden(name)
    m = make(size, base, ncls, multi, k, subtyping)
    h = synth.hash_uniform_torch(shape, seed, device=DEV)
    before = N.calls
    logits, y_prob, y_hat, a_raw, res, total = step(m, h, label, inst)
    assert N.calls >= before + 2 and "ClamTrainFn" in type(logits.grad_fn).__name__  # forward AND backward ran in the library
    assert md(logits, g["logits"]) < TOL and md(a_raw, g["A_raw"]) < TOL and md(res["features"], g["M"]) < TOL
    assert md(y_prob, g["Y_prob"]) < TOL and np.array_equal(y_hat.cpu().numpy(), g["Y_hat"])
    assert abs(float(total) - float(g["loss"])) < TOL
    if inst:
        assert abs(float(res["instance_loss"]) - float(g["instance_loss"])) < TOL
        assert np.array_equal(res["inst_preds"], g["inst_preds"]) and np.array_equal(res["inst_labels"], g["inst_labels"])  # top-k ids right
    worst = 0.0
    for key, p in m.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        e = md(gr, g["grad." + key])
        worst = max(worst, e)
        assert gr.shape == p.shape and e < TOL, (key, e)
    print(f"{name}: max |grad - reference| = {worst:.1e}")


def _oracle_step(size, base, ncls, multi, h_cpu, label, k, inst, subtyping, masks=None, dropout=False):
    from oracle import torch_cpu as TO
    pn = synth.make_params_np(synth.clam_param_specs(size, n_classes=ncls, multi=multi, dropout=dropout), base)
    return TO.clam_train_step(h_cpu, pn, label, ncls, multi, k, inst, subtyping, masks=masks)


def test_train_step_with_dropout_masks_and_bag_gradient():
    """dropout 0.25 (the value the reference's scripts train with): the masks are drawn by torch's generator in the
    reference's order (after the ReLU, then inside attention_a, attention_b); the same masks go to the CPU oracle."""
    size, n = (192, 128, 64), 77
    m = make(size, 192, 2, False, 8, False, dropout=0.25)
    h = synth.hash_uniform_torch((n, 192), 31, device=DEV).requires_grad_(True)
    torch.manual_seed(1234)
    logits, _, _, a_raw, res, total = step(m, h, 1, True)
    torch.manual_seed(1234)
    ones = lambda c: torch.ones((n, c), device=DEV)
    masks = [F.dropout(ones(128), 0.25, True), F.dropout(ones(64), 0.25, True), F.dropout(ones(64), 0.25, True)]
    assert 0.15 < float((masks[0] == 0).float().mean()) < 0.35
    out, grads = _oracle_step(size, 192, 2, False, h.detach().cpu(), 1, 8, True, False, masks=[t.cpu() for t in masks], dropout=True)
    assert md(a_raw, out["A_raw"]) < TOL and md(logits, out["logits"]) < TOL and abs(float(total) - float(out["loss"])) < TOL
    for key, p in m.named_parameters():
        assert md(p.grad if p.grad is not None else torch.zeros_like(p), grads[key]) < TOL, key
    assert h.grad is not None and md(h.grad, grads["bag"]) < TOL  # d loss / d bag
    m.eval()  # eval: no masks, the inference kernels again
    with torch.no_grad():
        lg2 = m(h.detach())[0]
    assert lg2.grad_fn is None


@pytest.mark.parametrize("size,base,n,ncls,multi,k", [((1024, 512, 256), 1024, 300, 2, False, 8),   # CLAM's own default widths
                                                        ((192, 128, 64), 192, 5000, 2, False, 8),      # > 4096 rows: split weight reduction
                                                        ((384, 128, 64), 384, 33, 4, True, 4)])        # 4 branches
def test_train_step_vs_oracle_other_shapes(size, base, n, ncls, multi, k):
    m = make(size, base, ncls, multi, k, True)
    h = synth.hash_uniform_torch((n, size[0]), 55, device=DEV)
    logits, _, _, a_raw, res, total = step(m, h, ncls - 1, True)
    out, grads = _oracle_step(size, base, ncls, multi, h.cpu(), ncls - 1, k, True, True)
    assert md(a_raw, out["A_raw"]) < TOL and md(logits, out["logits"]) < TOL and abs(float(total) - float(out["loss"])) < TOL
    assert np.array_equal(res["inst_preds"], out["inst_preds"].numpy())
    for key, p in m.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        ref = grads[key]
        assert md(gr, ref) < max(TOL, 1e-5 * float(ref.abs().max())), (key, md(gr, ref))


def rel_l2(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_clam_mb_eval_forward_vs_reference_golden():
    g = golden("clam_mb_hipt_big_n333")
    m = make((192, 128, 64), 193, 3, True, 8, True).eval()
    h = synth.hash_uniform_torch((333, 192), 25, device=DEV)
    before = N.calls
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
        att = m(h, attention_only=True)
    assert N.calls >= before + 2
    assert a_raw.shape == (3, 333) and md(a_raw, g["A_raw"]) < TOL and md(att, g["A_raw"]) < TOL
    assert md(logits, g["logits"]) < TOL and md(y_prob, g["Y_prob"]) < TOL and md(res["features"], g["M"]) < TOL
    assert np.array_equal(y_hat.cpu().numpy(), g["Y_hat"])


def test_clam_mb_inference_runs_the_streaming_kernels_100k_bf16():
    """CLAM_MB in eval mode (models/model_clam.py:226-264) on the kernels CLAM_SB's inference uses: bf16, 100 000 x 192, K = 3 branches.
    Round 6: ONE pass over the bag for all branches (hipt_clam_mb_forward: the streaming kernel forms the gate once per row and K logits and
    leaves h1 in HBM as bf16; a pooling kernel makes the K pooled vectors and logits) -- two launches, the bag read once -- beside the
    branch-by-branch form of round 5 (K launches of hipt_clam_sb_forward, `one_pass = False`: 76 us).  Against the module's own PyTorch-op
    forward in fp32 at CLAM_SB's bf16 bars (A_raw 4e-2, M rel-L2 2e-3), the two forms against each other, timed with the library's HIP events
    (one event pair around the two launches of a forward)."""
    m = make((192, 128, 64), 193, 3, True, 8, True).eval()
    h = synth.hash_uniform_torch((100_000, 192), 27, device=DEV)
    with torch.no_grad():
        ref = m._torch_forward(h, None, False, True, False)
        m.set_compute_dtype("bf16")
        m(h)  # (packs the weight images: K + 1 more native calls, once per set of weights)
        before = N.calls
        lg, yp, yh, a_raw, res = m(h, return_features=True)
        assert N.calls == before + 1 and lg.grad_fn is None  # ONE native call: hipt_clam_mb_forward
        att = m(h, attention_only=True)
        m.one_pass = False
        lg3, yp3, yh3, a_raw3, res3 = m(h, return_features=True)
        m.one_pass = True
        hb = h.bfloat16()
        times = {}
        for name, one in (("one pass", True), ("branch by branch", False)):
            m.one_pass = one
            for _ in range(3):
                m(hb)
            torch.cuda.synchronize()
            N.profile_enable(True)
            for _ in range(10):
                m(hb)
            torch.cuda.synchronize()
            pr = N.profile_read()
            times[name] = (pr["abmil_fused"][0] + pr.get("abmil_combine", (0.0, 0))[0], pr["abmil_fused"][1] + pr.get("abmil_combine", (0.0, 0))[1], pr)
            N.profile_enable(False)
        m.one_pass = True
    lr, ypr, yhr, ar, rr = ref
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    print(f"CLAM_MB bf16 100000x192, 3 branches vs fp32 PyTorch ops: A_raw max abs {md(a_raw, ar.cpu().numpy()):.2e}, M rel-L2 {rel(res['features'], rr['features']):.2e}, "
          f"logits max abs {md(lg, lr.cpu().numpy()):.2e}; one pass vs branch by branch: A_raw {float((a_raw - a_raw3).abs().max()):.2e}, M rel-L2 "
          f"{rel(res['features'], res3['features']):.2e}; HIP-event time per forward: {({k: round(v[0] / 10 * 1e3, 1) for k, v in times.items()})} us "
          f"(one pass: streaming kernel {times['one pass'][2]['abmil_fused'][0] / 10 * 1e3:.1f} + pooling kernel {times['one pass'][2]['abmil_combine'][0] / 10 * 1e3:.1f}, an event pair each)")
    assert a_raw.shape == (3, 100_000) and torch.equal(att, a_raw)
    assert md(a_raw, ar.cpu().numpy()) < 4e-2 and rel(res["features"], rr["features"]) < 2e-3 and md(lg, lr.cpu().numpy()) < 2e-3
    assert float((a_raw - a_raw3).abs().max()) < 1e-4 and rel(res["features"], res3["features"]) < 1e-3 and float((lg - lg3).abs().max()) < 1e-3
    assert int(yh) == int(yhr) == int(yh3) and abs(float(yp.sum()) - 1.0) < 1e-6
    assert times["one pass"][1] == 20 and times["branch by branch"][1] == 30, times  # (two launches a forward / three)
    assert times["one pass"][0] / 10 * 1e3 <= 50.0 and times["branch by branch"][0] / 10 * 1e3 <= 90.0, times


@pytest.mark.parametrize("n,K,s0", [(1, 2, 192), (31, 3, 192), (33, 4, 192), (257, 3, 384), (8 * 32 * 5 + 7, 2, 384), (70001, 3, 192)])
def test_clam_mb_one_pass_ragged_bags_and_branch_counts(n, K, s0):
    """hipt_clam_mb_forward's edges: bags of less than a block, a block +- a row, more blocks than the pooling kernel's prefetch depth, 2 / 3 / 4
    branches, both bag widths -- against an fp64 evaluation on the bf16-rounded operands (bars as test_clam_stream_kernel_ragged_bags...)."""
    m = make((s0, 128, 64), 190 + K, K, True, 1, True).eval().set_compute_dtype("bf16")
    h = synth.hash_uniform_torch((n, s0), 170 + n % 97, device=DEV)
    with torch.no_grad():
        m(h)  # (packs)
        before = N.calls
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
    assert N.calls == before + 1
    p = {k: v.detach().float().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    r16 = lambda t: torch.from_numpy(t).bfloat16().double().numpy()
    x = h.bfloat16().double().cpu().numpy()
    h1 = np.maximum(x @ r16(p["attention_net.0.weight"]).T + p["attention_net.0.bias"], 0)
    g = np.tanh(r16(h1) @ r16(p["attention_net.2.attention_a.0.weight"]).T + p["attention_net.2.attention_a.0.bias"]) * \
        (1 / (1 + np.exp(-(r16(h1) @ r16(p["attention_net.2.attention_b.0.weight"]).T + p["attention_net.2.attention_b.0.bias"]))))
    A = g @ p["attention_net.2.attention_c.weight"].T + p["attention_net.2.attention_c.bias"]  # [n, K]
    for k in range(K):
        pw = np.exp(A[:, k] - A[:, k].max())
        Mk = (pw / pw.sum()) @ r16(h1)   # (the pooling kernel reads h1 as the first pass rounded it to bf16)
        lk = Mk @ p[f"classifiers.{k}.weight"][0] + p[f"classifiers.{k}.bias"][0]
        assert md(a_raw[k], A[:, k]) < 4e-2, (n, k)
        assert rel_l2(res["features"][k], Mk) < 3e-3 and abs(float(logits[0, k]) - lk) < 2e-3, (n, k)
    assert bool(torch.isfinite(logits).all()) and abs(float(y_prob.sum()) - 1) < 1e-6


def test_clam_mb_long_bag_pools_over_many_workgroups():
    """A slide-sized bag through CLAM_MB in fp32 with the instance branch's top-k (eval mode: the inference kernels, branch by
    branch, ids by hipt_topk_rows, the selected h1 rows recomputed), against the module's own PyTorch-op forward on the same device, and
    timed.  (Round 4 ran this on the training kernels' forward, whose pooling of long bags spreads over workgroups with fp32 atomics:
    test_train_step_vs_oracle_other_shapes still covers that path at 5 000 rows.)"""
    import time
    m = make((192, 128, 64), 193, 3, True, 8, True).eval()
    h = synth.hash_uniform_torch((60000, 192), 26, device=DEV)
    label = torch.tensor([1], device=DEV)
    with torch.no_grad():
        ref = m._torch_forward(h, label, True, True, False)
        got = m(h, label=label, instance_eval=True, return_features=True)
        for _ in range(3):
            m(h, return_features=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m(h, return_features=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
    print(f"CLAM_MB fp32 forward, 60000 x 192, 3 branches: {dt * 1e3:.2f} ms per call")
    lg, yp, yh, a_raw, res = got
    lr, ypr, yhr, ar, rr = ref
    assert md(a_raw, ar.cpu().numpy()) < 1e-4 and md(lg, lr.cpu().numpy()) < 1e-4 and md(yp, ypr.cpu().numpy()) < 1e-5
    assert md(res["features"], rr["features"].cpu().numpy()) < 1e-4 and int(yh) == int(yhr)
    assert abs(float(res["instance_loss"]) - float(rr["instance_loss"])) < 1e-4
    assert dt < 3e-3


def test_topk_rows_on_device_matches_torch_topk():
    a = synth.hash_uniform_torch((3, 1000), 9, device=DEV)
    a[1, 17] = a[1, 500]  # a tie: lowest index first
    ids = torch.empty((3, 2, 8), dtype=torch.int64, device=DEV)
    N.call("hipt_topk_rows", N.ptr(a), 3, 1000, 8, N.ptr(ids), N.stream_ptr(a.device))
    ref_p = torch.topk(a, 8, dim=1)[1]
    ref_n = torch.topk(-a, 8, dim=1)[1]
    for r in (0, 2):
        assert torch.equal(ids[r, 0], ref_p[r]) and torch.equal(ids[r, 1], ref_n[r])
    assert torch.equal(a[1].gather(0, ids[1, 0]), torch.topk(a[1], 8)[0])  # same VALUES whatever the order among equals
    with pytest.raises(RuntimeError, match="exceeds"):
        N.call("hipt_topk_rows", N.ptr(a), 3, 4, 8, N.ptr(ids), N.stream_ptr(a.device))


def test_train_step_time_vs_pytorch_ops():
    """The launch-bound case the row exists for: a 100-row bag, forward + backward (printed; DESIGN.md quotes it)."""
    m = make((192, 128, 64), 192, 2, False, 8, False)
    h = synth.hash_uniform_torch((100, 192), 22, device=DEV)
    lab = torch.tensor([1], device=DEV)

    def hip():
        m.zero_grad(set_to_none=True)
        logits, _, _, _, res = m(h, label=lab, instance_eval=True)
        (0.7 * F.cross_entropy(logits, lab) + 0.3 * res["instance_loss"]).backward()

    def ops():
        m.zero_grad(set_to_none=True)
        logits, _, _, _, res = m._torch_forward(h, lab, True, False, False)
        (0.7 * F.cross_entropy(logits, lab) + 0.3 * res["instance_loss"]).backward()

    def clock(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    t_hip, t_ops = clock(hip), clock(ops)
    print(f"CLAM_SB hipt_big training step, N=100, instance_eval: HIP kernels {t_hip:.0f} us, PyTorch-op path {t_ops:.0f} us per step (host-inclusive)")
    assert t_hip < 1.5 * t_ops  # (both are host-bound at this size; the kernels must at least not lose)
