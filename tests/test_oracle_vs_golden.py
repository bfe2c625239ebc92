"""Pins the numpy oracle against outputs of the reference's own modules (tests/golden/*.npz,
made by tests/golden/make_golden.py in the build container).  CPU only."""
import numpy as np
import pytest

from conftest import golden
from hipt_abmil_atec23_amd import synth
from oracle import hipt_oracle as O

TOL = 1e-4  # BASELINE.json north_star: "within 1e-4 fp32"
ROWS = [0, 1, 128, 256]


def maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def P(specs, base):
    return synth.make_params_np(specs, base)


def test_hash_numpy_equals_torch():
    import torch
    a = synth.hash_uniform_np((3, 1000), 5, 0.3, 0.7)
    b = synth.hash_uniform_torch((3, 1000), 5, 0.3, 0.7).numpy()
    assert np.array_equal(a, b)
    assert a.dtype == np.float32 and abs(float(a.mean()) - 0.7) < 0.02
    assert synth.hash_uniform_np((10,), 1).min() >= -1.0 and synth.hash_uniform_np((10,), 1).max() < 1.0


def test_vit256_full_config():
    g = golden("vit256_full")
    p = P(synth.vit_param_specs("vit256"), 256)
    x = synth.hash_uniform_np((2, 3, 256, 256), 2)
    tok = O.vit256_prepare_tokens(x, p)
    assert maxdiff(tok[:, ROWS], g["tokens_rows"]) < TOL
    assert maxdiff(O.interpolate_pos_encoding(p["pos_embed"], 256, 256, 256, 16), g["pos"]) < 1e-5
    t = tok
    for i in range(12):
        t = O.block(t, p, i, 6)
        if i in (0, 5, 11):
            assert maxdiff(t[:, ROWS], g[f"blk{i}_rows"]) < TOL, i
    out = O.layer_norm(t, p["norm.weight"], p["norm.bias"])[:, 0]
    assert maxdiff(out, g["out"]) < TOL
    attn = O.vit_last_selfattention(tok, p, 6)
    assert maxdiff(attn[:, :, 0], g["attn_cls"]) < 1e-5
    assert maxdiff(attn[:, :, 200], g["attn_row200"]) < 1e-5
    # the synthetic weights must give a softmax that is far from uniform (1/257 = 0.0039)
    assert g["attn_cls"].max() > 0.02


def test_vit_reduced_nonsquare():
    g = golden("vit_small_cfg")
    p = P(synth.vit_param_specs("vit256", embed_dim=64, depth=2, num_heads=2), 64)
    x = synth.hash_uniform_np((2, 3, 64, 96), 22)
    tok = O.vit256_prepare_tokens(x, p)
    assert maxdiff(tok, g["tokens"]) < 1e-5
    assert maxdiff(O.interpolate_pos_encoding(p["pos_embed"], 24, 64, 96, 16), g["pos"]) < 1e-6
    assert maxdiff(O.vit256_forward(x, p, 2), g["out"]) < 1e-5
    assert maxdiff(O.vit_last_selfattention(tok, p, 2), g["attn"]) < 1e-6
    assert maxdiff(np.stack(O.vit_intermediate_layers(tok, p, 2, n=2)), g["inter"]) < 1e-5


def test_vit4k():
    g = golden("vit4k")
    p = P(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096)
    g16 = synth.hash_uniform_np((1, 384, 16, 16), 4)
    g34 = synth.hash_uniform_np((2, 384, 3, 4), 44)
    tok = O.vit4k_prepare_tokens(g16, p)
    assert maxdiff(tok[:, ROWS], g["tokens16_rows"]) < 1e-5
    assert maxdiff(O.interpolate_pos_encoding(p["pos_embed"], 256, 16, 16, 1), g["pos16"]) < 1e-6
    assert maxdiff(O.interpolate_pos_encoding(p["pos_embed"], 12, 3, 4, 1), g["pos34"]) < 1e-6
    assert maxdiff(O.vit4k_forward(g16, p), g["out16"]) < TOL
    assert maxdiff(O.vit4k_forward(g34, p), g["out34"]) < TOL
    assert maxdiff(O.vit_last_selfattention(tok, p, 6)[:, :, 0], g["attn_cls16"]) < 1e-5


def test_hipt4k_composite_small_region():
    g = golden("hipt4k_1024x768")
    p256 = P(synth.vit_param_specs("vit256"), 256)
    p4k = P(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096)
    x = synth.hash_uniform_np((1, 3, 1024, 768), 3)
    out, f = O.hipt4k_forward(x, p256, p4k, return_cls256=True)
    assert maxdiff(f, g["cls256"]) < TOL
    assert maxdiff(out, g["out"]) < TOL


def test_region_attention_scores():
    """HIPT_4K._get_region_attention_scores (hipt_4k.py:135-160), tensor half, against the reference ViTs' own maps."""
    g = golden("hipt4k_attn_1024x768_s4")
    p256 = P(synth.vit_param_specs("vit256"), 256)
    p4k = P(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096)
    x = synth.hash_uniform_np((1, 3, 1024, 768), 3)
    b, a256, a4k = O.region_attention_scores(x, p256, p4k, scale=4)
    assert a256.shape == g["attention_256"].shape == (12, 6, 64, 64) and a4k.shape == g["attention_4k"].shape == (6, 256, 192)
    assert maxdiff(a256, g["attention_256"]) < 1e-6 and maxdiff(a4k, g["attention_4k"]) < 1e-6
    assert np.array_equal(((b.transpose(0, 2, 3, 1) + 1) / 2.0 * 255.0).astype(np.uint8), g["patches_u8"])  # index work: bit-exact


def test_patchify_and_grid_order_bit_exact():
    """Index work is bit-exact: patch k = p1*h_256 + p2 and grid[0,:,i,j] = f[i*h_256+j]."""
    img = np.arange(1 * 2 * 512 * 768, dtype=np.float32).reshape(1, 2, 512, 768)
    b = O.patchify_256(img, 2, 3)
    assert b.shape == (6, 2, 256, 256)
    for p1 in range(2):
        for p2 in range(3):
            assert np.array_equal(b[p1 * 3 + p2], img[0, :, p1 * 256:(p1 + 1) * 256, p2 * 256:(p2 + 1) * 256])
    f = np.arange(6 * 5, dtype=np.float32).reshape(6, 5)
    grid = O.cls_grid(f, 2, 3)
    assert grid.shape == (1, 5, 2, 3)
    for i in range(2):
        for j in range(3):
            assert np.array_equal(grid[0, :, i, j], f[i * 3 + j])
    crop, w, h = O.prepare_img_tensor(np.zeros((1, 3, 600, 1000), np.float32))
    assert crop.shape == (1, 3, 512, 768) and (w, h) == (2, 3)


CLAM_CASES = [
    ("clam_384_n2000", (384, 128, 64), 384, (2000, 384), 1, 1, 8, False, False),
    ("clam_384_n777", (384, 128, 64), 384, (777, 384), 11, 0, 8, False, False),
    ("clam_384_n1", (384, 128, 64), 384, (1, 384), 12, None, 8, False, False),
    ("clam_hipt_big_n500", (192, 128, 64), 192, (500, 192), 5, 1, 8, False, False),
    ("clam_hipt_smallest_n100", (192, 8, 4), 8, (100, 192), 6, 0, 4, True, False),
    ("clam_small_dropout_n300", (1024, 512, 256), 1024, (300, 1024), 7, None, 8, False, True),
]


@pytest.mark.parametrize("name,size,base,shape,seed,label,k,subtyping,dropout", CLAM_CASES)
def test_clam_sb(name, size, base, shape, seed, label, k, subtyping, dropout):
    g = golden(name)
    p = P(synth.clam_param_specs(size, dropout=dropout), base)
    h = synth.hash_uniform_np(shape, seed)
    r = O.clam_sb_forward(h, p, k_sample=k, label=label, instance_eval=label is not None, subtyping=subtyping)
    assert maxdiff(r["A_raw"], g["A_raw"]) < TOL
    assert maxdiff(r["logits"], g["logits"]) < TOL
    assert maxdiff(r["Y_prob"], g["Y_prob"]) < TOL
    assert maxdiff(r["M"], g["M"]) < TOL
    assert np.array_equal(r["Y_hat"], g["Y_hat"])  # int64, bit-exact
    assert maxdiff(O.clam_sb_forward(h, p, attention_only=True), g["attention_only"]) < TOL
    if label is not None:
        ids = r["inst_ids"][0]
        assert np.array_equal(ids[:k], g["top_p"])  # indices bit-exact
        if len(ids) > k:
            assert np.array_equal(ids[k:], g["top_n"])
        assert abs(r["instance_loss"] - float(g["instance_loss"])) < TOL
        preds = np.concatenate([np.argmax(l, axis=1) for l in r["inst_logits"]])
        assert np.array_equal(preds, g["inst_preds"])


def test_attn_net_gated():
    g = golden("attn_net_gated_384_256")
    spec = {"attention_a.0.weight": ((256, 384), 0.09, 0.0), "attention_a.0.bias": ((256,), 0.02, 0.0),
            "attention_b.0.weight": ((256, 384), 0.09, 0.0), "attention_b.0.bias": ((256,), 0.02, 0.0),
            "attention_c.weight": ((1, 256), 0.3, 0.0), "attention_c.bias": ((1,), 0.02, 0.0)}
    p = P(spec, 9)
    h = synth.hash_uniform_np((2000, 384), 1)
    A, x = O.attn_net_gated(h, p)
    assert x is h
    assert maxdiff(A, g["A"]) < TOL


def test_fp64_oracle_agrees_with_fp32_reference():
    """fp64 evaluation of the oracle stays within the fp32 tolerance of the reference output."""
    g = golden("vit4k")
    p = {k: v.astype(np.float64) for k, v in P(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096).items()}
    out = O.vit4k_forward(synth.hash_uniform_np((1, 384, 16, 16), 4).astype(np.float64), p)
    assert maxdiff(out, g["out16"]) < TOL


# ---------------------------------------------------------------------------------------------
# the PyTorch-CPU form of the oracle (oracle/torch_cpu.py: what bench.py times as the CPU baseline)
# ---------------------------------------------------------------------------------------------
def _tp(specs, base):
    from oracle import torch_cpu as TO
    return TO.to_torch(P(specs, base))


def test_torch_cpu_oracle_vit256_vit4k_hipt4k():
    import torch
    from oracle import torch_cpu as TO
    s256, s4k = synth.vit_param_specs("vit256"), synth.vit_param_specs("vit4k", embed_dim=192, depth=6)
    p256n, p4kn = P(s256, 256), P(s4k, 4096)
    p256, p4k = TO.to_torch(p256n), TO.to_torch(p4kn)
    pos256 = torch.from_numpy(O.interpolate_pos_encoding(p256n["pos_embed"], 256, 256, 256, 16))
    with torch.no_grad():
        g = golden("vit256_full")
        x = torch.from_numpy(synth.hash_uniform_np((2, 3, 256, 256), 2))
        assert maxdiff(TO.vit256_forward(x, p256, pos256).numpy(), g["out"]) < TOL
        g = golden("vit4k")
        pos16 = torch.from_numpy(O.interpolate_pos_encoding(p4kn["pos_embed"], 256, 16, 16, 1))
        pos34 = torch.from_numpy(O.interpolate_pos_encoding(p4kn["pos_embed"], 12, 3, 4, 1))
        assert maxdiff(TO.vit4k_forward(torch.from_numpy(synth.hash_uniform_np((1, 384, 16, 16), 4)), p4k, pos16).numpy(), g["out16"]) < TOL
        assert maxdiff(TO.vit4k_forward(torch.from_numpy(synth.hash_uniform_np((2, 384, 3, 4), 44)), p4k, pos34).numpy(), g["out34"]) < TOL
        g = golden("hipt4k_1024x768")  # non-square: pins the patch order and the grid permute of the torch form too
        x = torch.from_numpy(synth.hash_uniform_np((1, 3, 1024, 768), 3))
        pos43 = torch.from_numpy(O.interpolate_pos_encoding(p4kn["pos_embed"], 12, 4, 3, 1))  # 1024 x 768 -> a 4 x 3 [CLS] grid
        assert maxdiff(TO.hipt4k_forward(x, p256, p4k, pos256, pos43).numpy(), g["out"]) < TOL


@pytest.mark.parametrize("name,size,base,shape,seed", [("clam_384_n2000", (384, 128, 64), 384, (2000, 384), 1),
                                                        ("clam_hipt_big_n500", (192, 128, 64), 192, (500, 192), 5),
                                                        ("clam_384_n1", (384, 128, 64), 384, (1, 384), 12)])
def test_torch_cpu_oracle_clam_sb(name, size, base, shape, seed):
    import torch
    from oracle import torch_cpu as TO
    g = golden(name)
    p = _tp(synth.clam_param_specs(size), base)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, M = TO.clam_sb_forward(torch.from_numpy(synth.hash_uniform_np(shape, seed)), p)
    assert maxdiff(a_raw.numpy(), g["A_raw"]) < TOL and maxdiff(logits.numpy(), g["logits"]) < TOL
    assert maxdiff(y_prob.numpy(), g["Y_prob"]) < TOL and maxdiff(M.numpy(), g["M"]) < TOL
    assert np.array_equal(y_hat.numpy(), g["Y_hat"])


# ---------------------------------------------------------------------------------------------
# CLAM training step: the reference's gradients (tests/golden/clam_*grad*.npz) pin the torch-CPU oracle's autograd step
# and the host mirror's own CPU training path
# ---------------------------------------------------------------------------------------------
TRAIN_CASES = [
    # name, size, base, (N, S0) seed, label, n_classes, multi, k, subtyping, instance_eval
    ("clam_grad_hipt_big_n15", (192, 128, 64), 192, (15, 192), 21, 1, 2, False, 8, False, True),
    ("clam_grad_hipt_big_n100", (192, 128, 64), 192, (100, 192), 22, 1, 2, False, 8, False, True),
    ("clam_grad_hipt_big_n2000", (192, 128, 64), 192, (2000, 192), 23, 1, 2, False, 8, False, True),
    ("clam_grad_hipt_big_n100_bagonly", (192, 128, 64), 192, (100, 192), 22, 0, 2, False, 8, False, False),
    ("clam_grad_hipt_smallest_n100", (192, 8, 4), 8, (100, 192), 6, 0, 2, False, 4, True, True),
    ("clam_mb_grad_hipt_big_n100", (192, 128, 64), 193, (100, 192), 24, 2, 3, True, 8, True, True),
]


@pytest.mark.parametrize("name,size,base,shape,seed,label,ncls,multi,k,subtyping,inst", TRAIN_CASES)
def test_torch_cpu_oracle_train_step_gradients(name, size, base, shape, seed, label, ncls, multi, k, subtyping, inst):
    import torch
    from oracle import torch_cpu as TO
    g = golden(name)
    pn = P(synth.clam_param_specs(size, n_classes=ncls, multi=multi), base)
    out, grads = TO.clam_train_step(torch.from_numpy(synth.hash_uniform_np(shape, seed)), pn, label, ncls, multi, k, inst, subtyping)
    assert maxdiff(out["logits"].numpy(), g["logits"]) < 1e-5 and maxdiff(out["A_raw"].numpy(), g["A_raw"]) < 1e-5
    assert abs(float(out["loss"]) - float(g["loss"])) < 1e-5
    for key, gr in grads.items():
        if key != "bag":
            assert maxdiff(gr.numpy(), g["grad." + key]) < 1e-5, key


@pytest.mark.parametrize("name,size,base,shape,seed,label,ncls,multi,k,subtyping,inst", TRAIN_CASES[1::2])
def test_host_mirror_cpu_training_path_gradients(name, size, base, shape, seed, label, ncls, multi, k, subtyping, inst):
    """The package's CLAM_SB / CLAM_MB on CPU tensors (PyTorch ops: main.py still trains where there is no GPU)."""
    import torch
    from hipt_abmil_atec23_amd import CLAM_MB, CLAM_SB
    g = golden(name)
    m = (CLAM_MB if multi else CLAM_SB)(size_arg=list(size), k_sample=k, n_classes=ncls, subtyping=subtyping)
    m.load_state_dict(synth.make_state_dict(synth.clam_param_specs(size, n_classes=ncls, multi=multi), base))
    m.train()
    lab = torch.tensor([label])
    logits, _, _, a_raw, res = m(torch.from_numpy(synth.hash_uniform_np(shape, seed)), label=lab, instance_eval=inst)
    loss = torch.nn.functional.cross_entropy(logits, lab)
    (0.7 * loss + 0.3 * res["instance_loss"] if inst else loss).backward()
    assert maxdiff(a_raw.detach().numpy(), g["A_raw"]) < 1e-5
    for key, p in m.named_parameters():
        assert maxdiff((p.grad if p.grad is not None else torch.zeros_like(p)).numpy(), g["grad." + key]) < 1e-5, key


# ---- the OUTLIER weight family (round 6): LayerNorm gains 0.05 ... 20, residual channels of magnitude 100, logits up to +-120 ----
def test_outlier_weights_numpy_equals_torch():
    a = synth.make_vit_outlier_params_np(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096, 6)
    b = synth.make_vit_outlier_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096, 6)
    assert all(np.array_equal(a[k], b[k].numpy()) for k in a)
    assert np.abs(a["phi.0.bias"]).max() == 100.0 and a["blocks.2.norm1.weight"].max() <= 20.0


def test_vit256_outlier_family():
    g = golden("vit256_outlier")
    p = synth.make_vit_outlier_params_np(synth.vit_param_specs("vit256"), 256, 6)
    x = synth.hash_uniform_np((2, 3, 256, 256), 2)
    tok = O.vit256_prepare_tokens(x, p)
    scale = float(np.abs(g["blk11_rows"]).max())  # the residual stream carries |x| ~ 100: fp32 round-off scales with it
    assert 90 < scale < 120 and g["logit_absmax_per_block"].max() > 100  # the fixture IS a stress case
    assert maxdiff(tok[:, ROWS], g["tokens_rows"]) < TOL * scale
    t = tok
    for i in range(12):
        t = O.block(t, p, i, 6)
        if i in (0, 5, 11):
            assert maxdiff(t[:, ROWS], g[f"blk{i}_rows"]) < TOL * scale, i
    out = O.layer_norm(t, p["norm.weight"], p["norm.bias"])[:, 0]
    assert maxdiff(out, g["out"]) < TOL * float(np.abs(g["out"]).max())
    assert maxdiff(O.vit_last_selfattention(tok, p, 6)[:, :, 0], g["attn_cls"]) < TOL


def test_hipt4k_outlier_family():
    g = golden("hipt4k_outlier_1024")
    p256 = synth.make_vit_outlier_params_np(synth.vit_param_specs("vit256"), 256, 6)
    p4k = synth.make_vit_outlier_params_np(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096, 6)
    x = synth.hash_uniform_np((1, 3, 1024, 768), 3)
    out, f = O.hipt4k_forward(x, p256, p4k, return_cls256=True)
    assert maxdiff(f, g["cls256"]) < TOL * float(np.abs(g["cls256"]).max())
    assert maxdiff(out, g["out"]) < TOL * float(np.abs(g["out"]).max())


@pytest.mark.parametrize("tag,bound", [("lo", 50.0), ("hi", 70.0)])
def test_clam_outlier_bounds(tag, bound):
    g = golden("clam_outlier_n2000")
    p = synth.scale_clam_attention_c_np(synth.make_params_np(synth.clam_param_specs((384, 128, 64)), 384), bound)
    assert abs(float(g[f"{tag}_bound"]) - bound) < 1e-3
    r = O.clam_sb_forward(synth.hash_uniform_np((2000, 384), 1), p)
    assert maxdiff(r["A_raw"], g[f"{tag}_A_raw"]) < TOL and maxdiff(r["M"], g[f"{tag}_M"]) < TOL
    assert maxdiff(r["logits"], g[f"{tag}_logits"]) < TOL and np.array_equal(np.asarray(r["Y_hat"]).ravel(), g[f"{tag}_Y_hat"].ravel())
