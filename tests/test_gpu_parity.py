"""GPU parity tests (-m gpu): the HIP path (through the C ABI) against the numpy oracle on the same
seeded inputs and against the golden vectors captured from the reference modules.

Tolerances (BASELINE.json north_star): fp32 mode 1e-4 absolute on attention weights / logits /
features, indices bit-exact.  bf16 mode cannot meet 1e-4 (CPU bf16 autocast of the
reference itself is 2.4e-2 max-abs off); SURVEY.md 8d recommends rel-L2 <= 2e-2 and cosine >= 0.999 vs the fp32
golden; the tests hold the build to twice what it measures (rel-L2 <= 1.3e-2, cosine >= 0.9999), stated and printed per test.
"""
import os
from functools import partial

import numpy as np
import pytest
import torch

from conftest import golden
from hipt_abmil_atec23_amd import _native as N
from hipt_abmil_atec23_amd import functional as Fn
from hipt_abmil_atec23_amd import synth
from oracle import hipt_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda:0"
ROWS = [0, 1, 128, 256]


def md(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.max(np.abs(a.astype(np.float64) - np.asarray(b, np.float64))))


def rel_l2(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def cosine(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def vit256():
    from hipt_abmil_atec23_amd.vision_transformer import vit_small
    m = vit_small(patch_size=16, num_classes=0)
    m.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
    return m.eval().to(DEV)


@pytest.fixture(scope="module")
def vit4k():
    from hipt_abmil_atec23_amd.vision_transformer4k import vit4k_xs
    m = vit4k_xs(num_classes=0)
    m.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
    return m.eval().to(DEV)


def test_native_library_is_the_path():
    assert N.lib().hipt_abi_version() == N.ABI_VERSION
    before = N.calls
    x = synth.hash_uniform_torch((8, 64), 1, device=DEV)
    Fn.layernorm(x, torch.ones(64, device=DEV), torch.zeros(64, device=DEV))
    assert N.calls == before + 1


# ---------------------------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,D", [(1, 64), (257, 192), (1000, 384), (5, 1024)])
def test_layernorm(rows, D):
    x = synth.hash_uniform_np((rows, D), 31, 2.0, 0.3)
    w = synth.hash_uniform_np((D,), 32, 0.1, 1.0)
    b = synth.hash_uniform_np((D,), 33, 0.05)
    ref = O.layer_norm(x, w, b)
    out = Fn.layernorm(T(x), T(w), T(b), 1e-6)
    assert md(out, ref) < 1e-5
    out16 = Fn.layernorm(T(x), T(w), T(b), 1e-6, out_dtype=N.HIPT_BF16)
    assert out16.dtype == torch.bfloat16 and md(out16, ref) < 3e-2


# (M <= 1088 rows take the small-M kernel -- a wave per 16 x 32 output tile, operands straight from L2 -- larger ones the
#  sequence-tiled LDS kernel: both are covered)
@pytest.mark.parametrize("M,Nn,K", [(1, 64, 64), (257, 576, 192), (300, 384, 1536), (1000, 1152, 384), (130, 8, 192), (1088, 384, 384),
                                    (1089, 384, 384), (2056, 1152, 384)])
@pytest.mark.parametrize("mode", ["plain", "gelu", "relu", "resid"])
def test_linear_fp32(M, Nn, K, mode):
    a = synth.hash_uniform_np((M, K), 41)
    w = synth.hash_uniform_np((Nn, K), 42, 0.05)
    b = synth.hash_uniform_np((Nn,), 43, 0.1)
    r = synth.hash_uniform_np((M, Nn), 44)
    ref = O.linear(a.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    kw = {}
    if mode == "gelu":
        ref, kw = O.gelu(ref), dict(gelu=True)
    elif mode == "relu":
        ref, kw = np.maximum(ref, 0), dict(relu=True)
    elif mode == "resid":
        ref, kw = ref + r, dict(resid=T(r))
    out = Fn.linear(T(a), T(w), T(b), dtype=N.HIPT_F32, **kw)
    assert md(out, ref) < 2e-5


# ((100, 512, 1024): 32 k-steps, not a multiple of the small-M kernel's ring of six -- its guarded instantiation)
@pytest.mark.parametrize("M,Nn,K", [(257, 576, 192), (1000, 1152, 384), (300, 384, 1536), (1500, 384, 1536), (2056, 1152, 384), (100, 512, 1024)])
def test_linear_bf16(M, Nn, K):
    a = synth.hash_uniform_np((M, K), 41)
    w = synth.hash_uniform_np((Nn, K), 42, 0.05)
    b = synth.hash_uniform_np((Nn,), 43, 0.1)
    # reference on the bf16-rounded operands in fp64: isolates MFMA accumulation from input rounding
    a16 = T(a).bfloat16().float().cpu().numpy().astype(np.float64)
    w16 = T(w).bfloat16().float().cpu().numpy().astype(np.float64)
    ref = O.linear(a16, w16, b.astype(np.float64))
    out = Fn.linear(T(a), T(w), T(b), dtype=N.HIPT_BF16)
    assert md(out, ref) < 1e-4
    out16 = Fn.linear(T(a), T(w), T(b), dtype=N.HIPT_BF16, gelu=True, out_f32=False)
    assert out16.dtype == torch.bfloat16 and rel_l2(out16, O.gelu(ref)) < 5e-3


def test_linear_small_and_large_calls_agree():
    """The same rows through the small-M kernel (257 of them alone) and through the sequence-tiled kernel (as the head of 2056):
    two summation orders of the same products -- fp32 to 1e-5, and each row's result independent of the rest of ITS call."""
    a = T(synth.hash_uniform_np((2056, 384), 45))
    w, b = T(synth.hash_uniform_np((1152, 384), 46, 0.05)), T(synth.hash_uniform_np((1152,), 47, 0.1))
    big = Fn.linear(a, w, b, dtype=N.HIPT_F32)
    small = Fn.linear(a[:257], w, b, dtype=N.HIPT_F32)
    assert float((big[:257] - small).abs().max()) < 1e-5
    assert torch.equal(Fn.linear(a[100:357], w, b, dtype=N.HIPT_F32)[:157], small[100:])  # small kernel: rows do not see each other
    assert torch.equal(Fn.linear(a[:1500], w, b, dtype=N.HIPT_F32), big[:1500])            # nor in the tiled kernel


@pytest.mark.parametrize("B,ntok,heads,dh", [(2, 257, 6, 64), (3, 257, 6, 32), (2, 25, 2, 32), (1, 13, 6, 32), (2, 1, 2, 64),
                                               (1, 288, 3, 64)])
def test_attention_fp32(B, ntok, heads, dh):
    C = heads * dh
    qkv = synth.hash_uniform_np((B, ntok, 3 * C), 51, 1.5)
    scale = dh ** -0.5
    q, k, v = np.transpose(qkv.astype(np.float64).reshape(B, ntok, 3, heads, dh), (2, 0, 3, 1, 4))
    p = O.softmax(q @ k.transpose(0, 1, 3, 2) * scale, axis=-1)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, ntok, C)
    out, probs = Fn.attention(T(qkv), heads, scale, dtype=N.HIPT_F32, return_probs=True)
    assert md(probs, p) < 1e-5
    assert md(out, ref) < 2e-5
    out2, none = Fn.attention(T(qkv), heads, scale, dtype=N.HIPT_F32)
    assert none is None and torch.equal(out, out2)


def test_attention_rescue_large_logits():
    """a spiked key row forces max-subtraction to matter (exp overflow without it)"""
    B, ntok, heads, dh = 1, 257, 6, 64
    qkv = synth.hash_uniform_np((B, ntok, 3 * heads * dh), 52, 1.0)
    qkv[0, 100, heads * dh:2 * heads * dh] *= 60.0  # key 100 of every head
    scale = dh ** -0.5
    q, k, v = np.transpose(qkv.astype(np.float64).reshape(B, ntok, 3, heads, dh), (2, 0, 3, 1, 4))
    p = O.softmax(q @ k.transpose(0, 1, 3, 2) * scale, axis=-1)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, ntok, heads * dh)
    out, probs = Fn.attention(T(qkv), heads, scale, dtype=N.HIPT_F32, return_probs=True)
    assert torch.isfinite(out).all() and md(probs, p) < 1e-5 and md(out, ref) < 5e-5


@pytest.mark.parametrize("B,ntok,heads,dh", [(2, 257, 6, 64), (2, 257, 6, 32), (2, 25, 2, 32)])
def test_attention_bf16(B, ntok, heads, dh):
    C = heads * dh
    qkv = synth.hash_uniform_np((B, ntok, 3 * C), 51, 1.5)
    q16 = T(qkv).bfloat16().float().cpu().numpy().astype(np.float64)
    scale = dh ** -0.5
    q, k, v = np.transpose(q16.reshape(B, ntok, 3, heads, dh), (2, 0, 3, 1, 4))
    p = O.softmax(q @ k.transpose(0, 1, 3, 2) * scale, axis=-1)
    ref = (p @ v).transpose(0, 2, 1, 3).reshape(B, ntok, C)
    out, probs = Fn.attention(T(qkv), heads, scale, dtype=N.HIPT_BF16, return_probs=True)
    assert md(probs, p) < 1e-5  # probabilities are fp32 in both modes
    assert rel_l2(out, ref) < 1e-2  # P and O are rounded to bf16


# ---------------------------------------------------------------------------------------------
# ViT-256 / ViT-4K / HIPT_4K vs goldens from the reference
# ---------------------------------------------------------------------------------------------
def test_vit256_fp32_vs_reference_golden(vit256):
    g = golden("vit256_full")
    vit256.set_compute_dtype("fp32")
    x = synth.hash_uniform_torch((2, 3, 256, 256), 2, device=DEV)
    tok = vit256.prepare_tokens(x)
    assert md(tok[:, ROWS], g["tokens_rows"]) < TOL
    assert md(vit256.interpolate_pos_encoding(tok, 256, 256), g["pos"]) < 1e-5
    out = vit256(x)
    assert out.shape == (2, 384) and md(out, g["out"]) < TOL
    attn = vit256.get_last_selfattention(x)
    assert attn.shape == (2, 6, 257, 257)
    assert md(attn[:, :, 0], g["attn_cls"]) < TOL and md(attn[:, :, 200], g["attn_row200"]) < TOL
    # BASELINE config 2: a single 256x256 patch
    assert md(vit256(x[:1]), g["out"][:1]) < TOL
    # block-level API (Block.forward) agrees with the golden taps
    t = tok
    for i, blk in enumerate(vit256.blocks):
        t = blk(t)
        if i in (0, 5, 11):
            assert md(t[:, ROWS], g[f"blk{i}_rows"]) < TOL, i
    inter = vit256.get_intermediate_layers(x, n=1)
    assert md(inter[0][:, 0], g["out"]) < TOL


def test_vit256_bf16_vs_reference_golden(vit256):
    g = golden("vit256_full")
    vit256.set_compute_dtype("bf16")
    try:
        x = synth.hash_uniform_torch((2, 3, 256, 256), 2, device=DEV)
        out = vit256(x)
        print(f"ViT-256 bf16 vs reference: rel-L2 {rel_l2(out, g['out']):.2e}, cosine {cosine(out, g['out']):.6f}")
        assert rel_l2(out, g["out"]) < 1.3e-2 and cosine(out, g["out"]) > 0.9999  # 2 x measured (6.4e-3); SURVEY 8d allows 2e-2
        attn = vit256.get_last_selfattention(x)
        print(f"ViT-256 bf16 [CLS] attention row vs reference: max abs {md(attn[:, :, 0], g['attn_cls']):.2e}")
        assert md(attn[:, :, 0], g["attn_cls"]) < 1e-3  # 2 x measured (4.6e-4)
    finally:
        vit256.set_compute_dtype("fp32")


def test_vit_reduced_nonsquare_fp32():
    from functools import partial
    from hipt_abmil_atec23_amd.vision_transformer import VisionTransformer
    g = golden("vit_small_cfg")
    cfg = dict(embed_dim=64, depth=2, num_heads=2)
    m = VisionTransformer(patch_size=16, num_classes=0, mlp_ratio=4, qkv_bias=True,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **cfg)
    m.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256", **cfg), 64))
    m = m.eval().to(DEV)
    x = synth.hash_uniform_torch((2, 3, 64, 96), 22, device=DEV)
    assert md(m.prepare_tokens(x), g["tokens"]) < 1e-5
    assert md(m(x), g["out"]) < 1e-5
    assert md(m.get_last_selfattention(x), g["attn"]) < 1e-5
    assert md(torch.stack(m.get_intermediate_layers(x, n=2)), g["inter"]) < 1e-5
    assert md(m.patch_embed(x), O.patch_embed(x.cpu().numpy(), m.patch_embed.proj.weight.detach().cpu().numpy(),
                                              m.patch_embed.proj.bias.detach().cpu().numpy())) < 1e-5


def test_vit4k_vs_reference_golden(vit4k):
    g = golden("vit4k")
    vit4k.set_compute_dtype("fp32")
    g16 = synth.hash_uniform_torch((1, 384, 16, 16), 4, device=DEV)
    g34 = synth.hash_uniform_torch((2, 384, 3, 4), 44, device=DEV)
    assert md(vit4k.prepare_tokens(g16)[:, ROWS], g["tokens16_rows"]) < 1e-5
    assert md(vit4k(g16), g["out16"]) < TOL
    assert md(vit4k(g34), g["out34"]) < TOL
    assert md(vit4k.get_last_selfattention(g16)[:, :, 0], g["attn_cls16"]) < TOL
    vit4k.set_compute_dtype("bf16")
    try:
        o = vit4k(g16)
        print(f"ViT-4K bf16 vs reference: rel-L2 {rel_l2(o, g['out16']):.2e}")
        assert rel_l2(o, g["out16"]) < 1.2e-2 and cosine(o, g["out16"]) > 0.9999  # 2 x measured (5.5e-3)
    finally:
        vit4k.set_compute_dtype("fp32")


@pytest.fixture(scope="module")
def hipt():
    from hipt_abmil_atec23_amd import HIPT_4K
    m = HIPT_4K(None, None, DEV, DEV)
    m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
    m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
    return m.eval().to(DEV)


def test_hipt4k_small_region_fp32(hipt):
    g = golden("hipt4k_1024x768")
    hipt.set_compute_dtype("fp32")
    x = synth.hash_uniform_torch((1, 3, 1024, 768), 3, device=DEV)
    out = hipt(x)
    assert out.shape == (1, 192) and out.device.type == "cuda"
    assert md(out, g["out"]) < TOL
    d = hipt.forward_asset_dict(x)
    assert md(d["features_cls256"], g["cls256"]) < TOL  # patch order k = p1*h_256 + p2 (bit-exact index work)
    assert md(d["features_cls4k"], g["out"]) < TOL
    assert d["features_mean256_cls4k"].shape == (1, 576)
    # chunked ViT-256 passes give the same answer
    hipt.chunk = 5
    try:
        assert md(hipt(x), g["out"]) < TOL
    finally:
        hipt.chunk = 0
    # centre crop to multiples of 256 (hipt_4k.py:308-330): a padded region reduces to the same crop
    xp = torch.zeros((1, 3, 1024 + 100, 768 + 37), device=DEV)
    xp[:, :, 50:50 + 1024, 18:18 + 768] = x
    assert md(hipt(xp), g["out"]) < TOL


def test_vit256_bf16_is_batch_invariant_bitwise(vit256):
    """A patch's tokens must not depend on how many patches share the call (which row tile / MFMA fragment a row
    lands in): every layer output of 6 patches inside an 18-patch batch equals, bit for bit, that of the 6 alone."""
    x = synth.hash_uniform_torch((18, 3, 256, 256), 33, device=DEV)
    vit256.set_compute_dtype("bf16")
    try:
        a = vit256.get_intermediate_layers(x, n=12)
        b = vit256.get_intermediate_layers(x[12:18], n=12)
    finally:
        vit256.set_compute_dtype("fp32")
    for i, (p, q) in enumerate(zip(a, b)):
        assert torch.equal(p[12:18], q), f"layer {i}: max diff {float((p[12:18] - q).abs().max())}"


def test_cls_attention_row_vs_reference_golden(vit256, vit4k):
    """get_last_selfattention_cls = [:, :, 0, :] of the last block's attention map (all the heat-maps
    consume, hipt_4k.py:143-158) from the one-query kernels, never building [B, heads, N, N].  Checked against the
    REFERENCE's own maps (golden attn_cls / attn_cls16 = get_last_selfattention(x)[:, :, 0] of the imported modules):
    fp32 1e-4, bf16 5e-3 absolute on probabilities; rows sum to 1.  The full-map API is only a cross-check here."""
    g, g4 = golden("vit256_full"), golden("vit4k")
    x = synth.hash_uniform_torch((2, 3, 256, 256), 2, device=DEV)
    g16 = synth.hash_uniform_torch((1, 384, 16, 16), 4, device=DEV)
    before = N.calls
    vit256.set_compute_dtype("fp32")
    vit4k.set_compute_dtype("fp32")
    r256, r4k = vit256.get_last_selfattention_cls(x), vit4k.get_last_selfattention_cls(g16)
    assert N.calls > before
    assert r256.shape == (2, 6, 257) and r4k.shape == (1, 6, 257)
    e256, e4k = md(r256, g["attn_cls"]), md(r4k, g4["attn_cls16"])
    assert e256 < TOL and e4k < TOL, (e256, e4k)
    assert md(r256, vit256.get_last_selfattention(x)[:, :, 0].cpu().numpy()) < 1e-6  # same numbers as the full map
    try:
        vit256.set_compute_dtype("bf16")
        vit4k.set_compute_dtype("bf16")
        b256, b4k = vit256.get_last_selfattention_cls(x), vit4k.get_last_selfattention_cls(g16)
    finally:
        vit256.set_compute_dtype("fp32")
        vit4k.set_compute_dtype("fp32")
    eb256, eb4k = md(b256, g["attn_cls"]), md(b4k, g4["attn_cls16"])
    print(f"[CLS]-row attention vs reference: fp32 {e256:.1e} / {e4k:.1e}, bf16 {eb256:.1e} / {eb4k:.1e} (ViT-256 / ViT-4K)")
    assert eb256 < 5e-3 and eb4k < 5e-3
    for r in (r256, r4k, b256, b4k):
        assert float((r.sum(-1) - 1).abs().max()) < 1e-5


def test_vit256_bf16_cls_pruned_last_block_matches_full(vit256):
    """forward() runs the last block for the [CLS] query only (nothing else of it is consumed, vision_transformer.py:253);
    get_intermediate_layers() runs every block in full.  Same [CLS] feature up to the rounding of the two attention
    kernels (one-query fp32 dot products vs MFMA tiles with bf16 probabilities)."""
    x = synth.hash_uniform_torch((5, 3, 256, 256), 71, device=DEV)
    vit256.set_compute_dtype("bf16")
    try:
        pruned = vit256(x)
        full = vit256.get_intermediate_layers(x, n=1)[-1][:, 0]
    finally:
        vit256.set_compute_dtype("fp32")
    assert pruned.shape == (5, 384)
    assert rel_l2(pruned, full.cpu().numpy()) < 3e-3 and cosine(pruned, full.cpu().numpy()) > 0.99999


def test_vit256_streaming_kernels_vs_generic_kernels(vit256, monkeypatch):
    """The streaming kernels (packed weight images hipt_block_weights.*_pk made by hipt_vit_pack_weights, LayerNorm chaining,
    32x32x16 fused MLP with the 3-coefficient GELU, fused QKV + attention) against the generic kernels of the same operators
    reading the row-major matrices (HIPT_GENERIC=1: seqgemm.hip, attention.hip, mlp.hip with the erf GELU): other summation
    orders, the same model -- held to the bf16 bar, forward() and the full-block path alike."""
    x = synth.hash_uniform_torch((16, 3, 256, 256), 19, device=DEV)
    vit256.set_compute_dtype("bf16")
    try:
        pk = vit256._tokens(x)[0]
        assert all(pk.blocks[i].mlp_pk and pk.blocks[i].qkv_pk and pk.blocks[i].proj_pk and pk.blocks[i].qkv_att_pk for i in range(pk.w.depth))
        assert all(pk.blocks[i].mlp_pk_fmt == 3 for i in range(pk.w.depth))  # (16x16x32 MFMAs, the proj units in front)
        default = vit256(x), vit256.get_intermediate_layers(x, n=2)
        monkeypatch.setenv("HIPT_GENERIC", "1")
        generic = vit256(x), vit256.get_intermediate_layers(x, n=2)
    finally:
        monkeypatch.delenv("HIPT_GENERIC", raising=False)
        vit256.set_compute_dtype("fp32")
    rel = float((default[0] - generic[0]).norm() / generic[0].norm())
    rel_i = max(float((a - b).norm() / b.norm()) for a, b in zip(default[1], generic[1]))
    print(f"streaming kernels vs generic kernels: [CLS] features rel-L2 {rel:.2e}, last two blocks' tokens {rel_i:.2e}")
    assert 0 < rel < 1.3e-2 and rel_i < 1.3e-2  # (0 would mean the switch did nothing)
    # fp32 weights have no packed form: the size query says so and packing is refused
    pk32 = vit256._tokens(x)[0]
    assert N.lib().hipt_vit_packed_bytes(pk32.ref, N.PACK_MLP) == 0
    with pytest.raises(RuntimeError):
        N.call("hipt_vit_pack_weights", pk32.ref, 0, N.PACK_MLP, N.ptr(x), N.stream_ptr(x.device))


def test_vit256_fused_mlp_image_format_travels_with_the_image(vit256, monkeypatch):
    """The fused MLP's weight image is format 3 (csrc/mlp16.hip, 16x16x32 MFMAs, six units of the proj matrix in front: the output projection
    runs at the head of the MLP's tiles).  Bit-identical between batchings; HIPT_NO_PROJ_FOLD=1 runs proj as its own kernel from the same image
    (y1 rounded to bf16 on the way: the bf16 bar, not the bits); an image that claims another format (1: the 32x32x16 form retired in round 5) is
    not run through the streaming kernel -- the model takes the generic kernels and stays within the bf16 bar."""
    x = synth.hash_uniform_torch((16, 3, 256, 256), 23, device=DEV)
    vit256.set_compute_dtype("bf16")
    try:
        default = vit256(x)
        pk = vit256._tokens(x)[0]
        assert N.lib().hipt_vit_mlp_pack_format(pk.ref) == 3 and all(pk.blocks[i].mlp_pk_fmt == 3 and pk.blocks[i].mlp_pk for i in range(12))
        two = vit256(torch.cat([x, x]))      # 32 patches = whole 16-row fragments, like the 16: the same kernels, other tile positions
        sub = vit256(torch.cat([x[5:], x]))  # 27 patches: 27 * 257 rows are no whole fragments: row-major path
        monkeypatch.setenv("HIPT_NO_PROJ_FOLD", "1")
        unfolded = vit256(x)
        unfolded_full = vit256.get_intermediate_layers(x, n=1)[-1]
        monkeypatch.delenv("HIPT_NO_PROJ_FOLD")
        folded_full = vit256.get_intermediate_layers(x, n=1)[-1]
        for i in range(12):
            pk.blocks[i].mlp_pk_fmt = 1     # a stale image format
        stale = vit256(x)
    finally:
        vit256._packed.clear()
        vit256.set_compute_dtype("fp32")
    assert torch.equal(two[:16], default) and torch.equal(two[16:], default)
    assert float((sub[11:] - default).norm() / default.norm()) < 1.3e-2
    rel = float((stale - default).norm() / default.norm())
    rel_u = float((unfolded - default).norm() / default.norm())
    rel_f = float((unfolded_full - folded_full).norm() / folded_full.norm())
    print(f"stale image format -> generic kernels: [CLS] features rel-L2 {rel:.2e}; proj folded vs its own kernel: {rel_u:.2e} ([CLS]), {rel_f:.2e} (all tokens, row-major path)")
    assert 0 < rel < 1.3e-2 and 0 < rel_u < 5e-3 and 0 < rel_f < 5e-3


def test_vit_other_hidden_width_proj_fold_vs_oracle(monkeypatch):
    """The fused proj + MLP kernel at another hidden width (mlp_ratio 2: 768 hidden units = 6 chunks, 30 ring units per tile pass instead of 54)
    and depth 3: bf16 against the fp32 numpy oracle at the bf16 bar, the folded and the unfolded form close to each other, and the image
    the library reports (format 3, D * D * 2 bytes more than the two MLP matrices)."""
    from hipt_abmil_atec23_amd.vision_transformer import VisionTransformer
    cfg = dict(embed_dim=384, depth=3, num_heads=6, mlp_ratio=2)
    m = VisionTransformer(patch_size=16, num_classes=0, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), **cfg)
    specs = synth.vit_param_specs("vit256", **cfg)
    m.load_state_dict(synth.make_state_dict(specs, 77))
    m = m.eval().to(DEV).set_compute_dtype("bf16")
    x = synth.hash_uniform_torch((16, 3, 256, 256), 78, device=DEV)
    pk = m._tokens(x)[0]
    assert N.lib().hipt_vit_mlp_pack_format(pk.ref) == 3
    assert N.lib().hipt_vit_packed_bytes(pk.ref, N.PACK_MLP) == 2 * 384 * 768 * 2 + 384 * 384 * 2
    folded = m(x)
    monkeypatch.setenv("HIPT_NO_PROJ_FOLD", "1")
    unfolded = m(x)
    monkeypatch.delenv("HIPT_NO_PROJ_FOLD")
    ref = O.vit256_forward(x.cpu().numpy(), synth.make_params_np(specs, 77), num_heads=6)
    r_f, r_u = rel_l2(folded, ref), rel_l2(unfolded, ref)
    r_fu = float((folded - unfolded).norm() / unfolded.norm())
    print(f"ViT D=384 depth 3 hidden 768, bf16 vs fp32 oracle: folded {r_f:.2e}, proj as its own kernel {r_u:.2e}; folded vs unfolded {r_fu:.2e}")
    assert r_f < 1.3e-2 and r_u < 1.3e-2 and 0 < r_fu < 1.3e-2  # (two bf16 paths, each ~8e-3 from the fp32 truth: measured 5.0e-3 apart)


def test_vit256_patch_embedding_from_fp32_pixels(vit256, monkeypatch):
    """csrc/embed32.hip reads the fp32 image itself (pixels rounded to bf16 in registers, weights through the LDS-DMA ring); the
    generic path (HIPT_GENERIC=1) makes a bf16 copy of the image and runs the generic GEMM over an im2col view.  Same bf16
    products, another summation order: the tokens agree with an fp64 Conv2d on the bf16-rounded operands to 1e-4 on both paths,
    also for a batch that is not a whole region and for a sub-batch (bitwise: a token's result does not depend on its
    neighbours; the sub-batch has 5 patches = 1285 rows so that both calls are on the streaming side of capi.hip's small_call()
    threshold of 1088 rows -- calls below it run the per-operator kernels, same values to the bf16 bar, other bits)."""
    x = synth.hash_uniform_torch((7, 3, 256, 256), 31, device=DEV)
    vit256.set_compute_dtype("bf16")
    try:
        fused = vit256(x)
        fused_sub = vit256(x[2:7])
        tok = vit256.prepare_tokens(x)
        monkeypatch.setenv("HIPT_GENERIC", "1")
        tok_plain = vit256.prepare_tokens(x)
    finally:
        monkeypatch.delenv("HIPT_GENERIC", raising=False)
        vit256.set_compute_dtype("fp32")
    # the tokens themselves against Conv2d on the bf16-rounded operands in fp64 (what both kernels compute, up to summation order)
    w = vit256.patch_embed.proj.weight.detach().to(torch.bfloat16).double()
    ref = torch.nn.functional.conv2d(x.to(torch.bfloat16).double(), w, vit256.patch_embed.proj.bias.detach().double(), stride=16)
    ref = ref.flatten(2).transpose(1, 2) + vit256.interpolate_pos_encoding(tok, 256, 256)[:, 1:].double()
    e_new, e_old = float((tok[:, 1:] - ref).abs().max()), float((tok_plain[:, 1:] - ref).abs().max())
    print(f"fused patch embedding: tokens vs fp64 Conv2d: max |err| {e_new:.2e} (generic path {e_old:.2e})")
    assert e_new < 1e-4 and e_old < 1e-4
    assert torch.equal(tok[:, 0], tok_plain[:, 0])
    assert torch.equal(fused[2:7], fused_sub)


def test_vit256_small_calls_agree_with_large_ones(vit256):
    """Calls of at most 1088 token rows (4 patches; capi.hip small_call()) run the per-operator kernels with the Linears on the
    wave-per-tile GEMM instead of the streaming kernels, whose fixed set-up costs more than such a call's work: the same features
    as the same patches inside a 7-patch call -- to 1e-4 in fp32, to the bf16 bar in bf16 -- and identical bits between two small
    calls that hold the same patch (rows do not meet on this path either)."""
    x = synth.hash_uniform_torch((7, 3, 256, 256), 41, device=DEV)
    for dt, bar in (("fp32", None), ("bf16", 1.3e-2)):
        vit256.set_compute_dtype(dt)
        try:
            big, small, one = vit256(x), vit256(x[2:5]), vit256(x[3:4])
        finally:
            vit256.set_compute_dtype("fp32")
        if bar is None:
            assert float((big[2:5] - small).abs().max()) < 1e-4
        else:
            assert float((big[2:5] - small).norm() / big[2:5].norm()) < bar
        assert torch.equal(small[1:2], one), dt


def test_hipt4k_patch_embedding_addresses_regions(hipt, monkeypatch):
    """The pixel-reading patch embedding inside HIPT_4K: 256 x 256 patches addressed inside non-square regions (grid 2 x 3, batch of
    two; one region per call takes the patch-range entry hipt_vit256_forward_range_px): the same features as the generic path
    (bf16 copy of the image + im2col GEMM + generic block kernels) up to the bf16 bar -- a patch read from the wrong place would
    be off by O(1) -- and identical bits for one region alone."""
    x = synth.hash_uniform_torch((2, 3, 512, 768), 37, device=DEV)
    hipt.set_compute_dtype("bf16")
    try:
        fused = hipt(x)
        fused_one = hipt(x[1:2])
        monkeypatch.setenv("HIPT_GENERIC", "1")
        plain = hipt(x)
    finally:
        monkeypatch.delenv("HIPT_GENERIC", raising=False)
        hipt.set_compute_dtype("fp32")
    rel = float((fused - plain).norm() / plain.norm())
    assert rel < 1.3e-2, rel
    assert torch.equal(fused[1:2], fused_one)


def test_vit256_activation_images_change_nothing(vit256, monkeypatch):
    """forward() keeps the residual stream, the attention-branch output and the pre-normalised QKV operands of blocks
    2..11 as fragment-blocked "activation images" (csrc/kernels.h) when the batch has whole 16-row fragments: a pure
    re-ordering of bytes in private buffers -- identical bits to the row-major path (HIPT_NO_IMG=1) when both sides run the
    same kernels (HIPT_NO_FUSED_ATTN=1: the fused QKV + attention kernel exists with images only; HIPT_NO_EMBED_LN=1: the patch
    embedding's own LayerNorm-1 -- another summation order than the LN-in-GEMM load -- exists with images only), also for a batch
    whose row count is not a multiple of 16 (which never uses them)."""
    vit256.set_compute_dtype("bf16")
    try:
        monkeypatch.setenv("HIPT_NO_FUSED_ATTN", "1")
        monkeypatch.setenv("HIPT_NO_EMBED_LN", "1")
        for nseq in (16, 48, 5):  # 16 * 257 and 48 * 257 rows: whole fragments; 5 * 257: not
            x = synth.hash_uniform_torch((nseq, 3, 256, 256), 23 + nseq, device=DEV)
            img = vit256(x)
            monkeypatch.setenv("HIPT_NO_IMG", "1")
            plain = vit256(x)
            monkeypatch.delenv("HIPT_NO_IMG")
            assert torch.equal(img, plain), nseq
    finally:
        monkeypatch.delenv("HIPT_NO_IMG", raising=False)
        monkeypatch.delenv("HIPT_NO_FUSED_ATTN", raising=False)
        monkeypatch.delenv("HIPT_NO_EMBED_LN", raising=False)
        vit256.set_compute_dtype("fp32")


def test_vit256_embedding_emits_first_block_operands(vit256, hipt, monkeypatch):
    """With activation images the patch embedding (embed32.hip, LNOUT) writes x as the fp32 image and LayerNorm-1 of the FIRST block as the bf16
    image, so that block 1 runs the fused QKV + attention kernel and the image-in fused MLP like blocks 2..11 (vision_transformer.py:235-246 then
    Block.forward :146-152).  Against the fp32 oracle at the bf16 bar of the other 12-block tests, against the path without it (HIPT_NO_EMBED_LN=1:
    LayerNorm in the QKV GEMM's load + the two-kernel attention) within that bar; a patch's features do not depend on its position in the call or
    on the call's size (bitwise); uint8 regions (planar and interleaved, 16 patches: whole fragments) give the bits of the normalised float path."""
    p = synth.make_params_np(synth.vit_param_specs("vit256"), 256)
    vit256.set_compute_dtype("bf16")
    try:
        x = synth.hash_uniform_torch((48, 3, 256, 256), 91, device=DEV)
        before = N.calls
        out = vit256(x)
        assert N.calls > before
        monkeypatch.setenv("HIPT_NO_EMBED_LN", "1")
        old = vit256(x)
        monkeypatch.delenv("HIPT_NO_EMBED_LN")
        ref = O.vit256_forward(x[:4].cpu().numpy(), p, num_heads=6)
        r_new, r_old, r_no = rel_l2(out[:4], ref), rel_l2(old[:4], ref), float((out - old).norm() / old.norm())
        print(f"ViT-256 bf16 vs fp32 oracle: embedding's LayerNorm {r_new:.2e}, LN-in-GEMM first block {r_old:.2e}; one vs the other {r_no:.2e}")
        assert r_new < 2e-2 and r_old < 2e-2 and 0 < r_no < 2e-2  # (two bf16 paths through 12 blocks)
        # position and call size: patches 16..31 alone, and the same patches at the head of a 32-patch call
        assert torch.equal(vit256(x[16:32]), out[16:32])
        assert torch.equal(vit256(torch.cat([x[16:32], x[:16]]))[:16], out[16:32])
    finally:
        monkeypatch.delenv("HIPT_NO_EMBED_LN", raising=False)
        vit256.set_compute_dtype("fp32")
    g = torch.Generator().manual_seed(7)
    u8 = torch.randint(0, 256, (1, 1024, 1024, 3), dtype=torch.uint8, generator=g)  # one decoded tile of 16 patches
    planar = u8.permute(0, 3, 1, 2).contiguous()
    ref_in = planar.float().div(255).sub(0.5).div(0.5)
    hipt.set_compute_dtype("bf16")
    try:
        want = hipt(ref_in.to(DEV))
        assert torch.equal(hipt(planar.to(DEV)), want) and torch.equal(hipt(u8.to(DEV)), want)
        monkeypatch.setenv("HIPT_NO_EMBED_LN", "1")
        assert not torch.equal(hipt(ref_in.to(DEV)), want)  # (the switch reaches this path: one region of 16 patches uses the images)
    finally:
        monkeypatch.delenv("HIPT_NO_EMBED_LN", raising=False)
        hipt.set_compute_dtype("fp32")


def _to_image(x):
    """[M, 384] row-major -> activation image (include/hipt_abmil.h, hipt_vit_attention_unit): fragment F, column chunk c, lane 16 g + i, 8 elements"""
    m = x.shape[0]
    return x.view(m // 16, 16, 12, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(m, 384)


def _from_image(img):
    m = img.shape[0]
    return img.view(m // 16, 12, 4, 16, 8).permute(0, 3, 1, 2, 4).contiguous().view(m, 384)


@pytest.mark.parametrize("nseq", [16, 48, 144, 528])  # (>= 128 patches: the one-wave-per-SIMD kernel; 528: two or three patches per workgroup)
def test_attention_unit_fused_kernel_vs_torch_and_two_kernels(vit256, nseq):
    """The attention unit of one block (qkv Linear + softmax(q k^T * scale) v, vision_transformer.py:121-128) through
    hipt_vit_attention_unit: the fused QKV + attention kernel, the QKV GEMM + attention kernel pair, and an fp32 PyTorch
    evaluation on the same bf16-rounded operands.  Bar: 1e-2 of the largest output (bf16 q / k / v / probabilities and a bf16
    result: measured 4e-3) against fp32, the two HIP paths within that of each other; every row of every patch is compared,
    the [CLS] row (which the fused kernel computes by another route) separately."""
    import ctypes as C
    vit256.set_compute_dtype("bf16")
    try:
        pk = vit256._tokens(synth.hash_uniform_torch((1, 3, 256, 256), 2, device=DEV))[0]
        blk = 3
        M = nseq * 257
        x = (synth.hash_uniform_torch((M, 384), 57 + nseq, device=DEV) * 2.0).bfloat16()  # (scores of std ~2: a peaky softmax, sensitive to any mis-ordered operand)
        xi = _to_image(x)
        need = N.lib().hipt_vit_workspace_bytes(pk.ref, nseq)
        ws = Fn.workspace(torch.device(DEV), need)
        outs = []
        for fused in (1, 0):
            o = torch.full((M, 384), float("nan"), dtype=torch.bfloat16, device=DEV)
            N.call("hipt_vit_attention_unit", pk.ref, blk, N.ptr(xi), nseq, N.ptr(o), fused, N.ptr(ws), ws.numel(), N.stream_ptr(torch.device(DEV)))
            torch.cuda.synchronize()
            outs.append(_from_image(o).float())
        att = vit256.blocks[blk].attn
        wq = att.qkv.weight.detach().bfloat16().float()
        qkv = (x.float() @ wq.t() + att.qkv.bias.detach().float()).bfloat16().float().view(nseq, 257, 3, 6, 64).permute(2, 0, 3, 1, 4)
        pr = torch.softmax((qkv[0] @ qkv[1].transpose(-1, -2)) * att.scale, dim=-1)
        ref = (pr @ qkv[2]).transpose(1, 2).reshape(M, 384)
    finally:
        vit256.set_compute_dtype("fp32")
    rmax = float(ref.abs().max())
    for name, o in zip(("fused", "two kernels"), outs):
        assert bool(torch.isfinite(o).all()), name
        e_all = float((o - ref).abs().max())
        e_cls = float((o.view(nseq, 257, 384)[:, 0] - ref.view(nseq, 257, 384)[:, 0]).abs().max())
        print(f"attention unit, {name}, {nseq} patches: max |err| vs fp32 torch {e_all:.2e} ([CLS] rows {e_cls:.2e}); |ref| max {rmax:.2f}")
        assert e_all < 1e-2 * rmax, name
    assert float((outs[0] - outs[1]).abs().max()) < 1e-2 * rmax


def test_vit256_fused_qkv_attention_vs_two_kernels(vit256, monkeypatch):
    """LayerNorm-chained blocks run the QKV projection INSIDE the attention kernel (csrc/qkv_attention.hip: q | k | v never reach
    HBM); HIPT_NO_FUSED_ATTN=1 runs the QKV GEMM and the attention as two kernels with the tensor between them.  Same bf16
    operands, other MFMA shapes and summation orders: the [CLS] features agree far inside the bf16 bar; bitwise against itself
    under another batching (a patch's result does not depend on its neighbours or on which workgroup takes it)."""
    x = synth.hash_uniform_torch((48, 3, 256, 256), 43, device=DEV)
    vit256.set_compute_dtype("bf16")
    try:
        before = N.calls
        fused = vit256(x)
        fused_sub = vit256(x[16:32])
        assert N.calls > before
        monkeypatch.setenv("HIPT_NO_FUSED_ATTN", "1")
        two = vit256(x)
    finally:
        monkeypatch.delenv("HIPT_NO_FUSED_ATTN", raising=False)
        vit256.set_compute_dtype("fp32")
    rel = float((fused - two).norm() / two.norm())
    print(f"fused QKV + attention vs QKV GEMM + attention kernel: [CLS] features rel-L2 {rel:.2e}")
    assert 0 < rel < 5e-3  # (0 would mean the switch did nothing)
    assert torch.equal(fused[16:32], fused_sub)


def test_hipt4k_region_batch_equals_single_regions(hipt):
    """R regions per call (throughput form) give the same features as R single-region calls."""
    x = synth.hash_uniform_torch((3, 3, 512, 768), 33, device=DEV)
    for dt, tol in (("fp32", 1e-5), ("bf16", 1e-5)):
        hipt.set_compute_dtype(dt)
        try:
            batched = hipt(x)
            singles = torch.cat([hipt(x[i:i + 1]) for i in range(3)], dim=0)
        finally:
            hipt.set_compute_dtype("fp32")
        assert batched.shape == (3, 192) and md(batched, singles.cpu().numpy()) < tol, dt
    # the same batch cut over two HIP streams (own workspace each): identical bits
    hipt.set_compute_dtype("bf16")
    try:
        hipt.streams = 1
        one = hipt(x)
        hipt.streams = 2
        two = hipt(x)
    finally:
        hipt.streams = 1
        hipt.set_compute_dtype("fp32")
    assert torch.equal(one, two)
    # the same with whole 16-row fragments per stream (16 patches x 257 rows each: activation images, packed weights)
    x2 = synth.hash_uniform_torch((2, 3, 1024, 1024), 34, device=DEV)
    hipt.set_compute_dtype("bf16")
    try:
        hipt.streams = 1
        one = hipt(x2)
        hipt.streams = 2
        two = hipt(x2)
    finally:
        hipt.streams = 1
        hipt.set_compute_dtype("fp32")
    assert torch.equal(one, two) and torch.equal(one[1:], two[1:])
    with pytest.raises(ValueError):
        hipt.forward_asset_dict(x)


def test_hipt4k_one_region_split_over_streams_by_patches(hipt):
    """Fewer regions than streams (the reference's batch of one): the call's PATCHES are spread over the streams
    (HIPT_4K._run_patch_split: image brought to the compute dtype once, hipt_vit256_forward_range per stream into the shared
    [CLS] grid, ViT-4K behind the join).  Same bits as one stream, for float and uint8 input, and for forward_asset_dict."""
    x = synth.hash_uniform_torch((1, 3, 2048, 2048), 35, device=DEV)  # 64 patches -> 2 x 32
    xu = ((x * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8)
    try:
        for dt in ("bf16", "fp32"):
            hipt.set_compute_dtype(dt)
            hipt.patch_streams = 1
            one, one_u = hipt(x), hipt(xu)
            d1 = hipt.forward_asset_dict(x)
            before = N.calls
            hipt.patch_streams = 2
            two, two_u = hipt(x), hipt(xu)
            d2 = hipt.forward_asset_dict(x)
            assert N.calls - before >= 3 * 3  # convert (bf16) / 2 ranges / ViT-4K per call
            assert torch.equal(one, two) and torch.equal(one_u, two_u), dt
            assert np.array_equal(d1["features_cls256"], d2["features_cls256"]) and np.array_equal(d1["features_cls4k"], d2["features_cls4k"])
        x3 = synth.hash_uniform_torch((1, 3, 2048, 3072), 36, device=DEV)  # 96 patches -> 32 / 32 / 32
        hipt.patch_streams = 1
        one3 = hipt(x3)
        hipt.patch_streams = 3
        assert torch.equal(hipt(x3), one3)
    finally:
        hipt.patch_streams = 1
        hipt.streams = 1
        hipt.set_compute_dtype("fp32")


def test_hipt4k_uint8_input_equals_normalised_float(hipt):
    """uint8 RGB regions (planar and interleaved) normalised on the device give the bits of the float path fed with
    torch's own ToTensor + Normalize(0.5, 0.5) arithmetic (hipt_model_utils.py:113-118)."""
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (2, 512, 768, 3), dtype=torch.uint8, generator=g)  # two decoded [W, H, RGB] tiles
    planar = u8.permute(0, 3, 1, 2).contiguous()
    ref_in = planar.float().div(255).sub(0.5).div(0.5)  # ToTensor then Normalize(mean 0.5, std 0.5)
    for dt in ("fp32", "bf16"):
        hipt.set_compute_dtype(dt)
        try:
            ref = hipt(ref_in.to(DEV))
            a = hipt(planar.to(DEV))
            b = hipt(u8.to(DEV))
        finally:
            hipt.set_compute_dtype("fp32")
        assert torch.equal(a, ref) and torch.equal(b, ref), dt
    # the normalisation alone, against torch on the CPU, bit for bit
    out = torch.empty((2, 3, 512, 768), dtype=torch.float32, device=DEV)
    src = u8.to(DEV)
    N.call("hipt_u8_normalize", N.ptr(src), 1, 2, 512 * 768, N.ptr(out), N.HIPT_F32, N.stream_ptr(torch.device(DEV)))
    assert torch.equal(out.cpu(), ref_in)
    # crop of a non-multiple-of-256 interleaved tile
    odd = torch.randint(0, 256, (1, 600, 300, 3), dtype=torch.uint8, generator=g)
    img, w, h = hipt.prepare_img_tensor(odd)
    assert img.shape == (1, 512, 256, 3) and (w, h) == (2, 1) and torch.equal(img, odd[:, 44:556, 22:278, :])


def test_extract_slide_to_feature_store_and_pool(hipt, tmp_path):
    """The steps either side of the path: regions -> HIPT_4K (batched, uint8) -> pt_files/{slide}.pt ->
    bag loader -> CLAM_SB, equal to calling the models directly."""
    from hipt_abmil_atec23_amd import CLAM_SB
    from hipt_abmil_atec23_amd.feature_store import extract_slide, load_bag
    g = torch.Generator().manual_seed(9)
    regions = torch.randint(0, 256, (5, 256, 512, 3), dtype=torch.uint8, generator=g)
    coords = torch.tensor([[4096 * i, 0] for i in range(5)])
    batches = [(regions[0:2].to(DEV), coords[0:2]), (regions[2:5].to(DEV), coords[2:5])]
    pt = extract_slide(hipt, batches, str(tmp_path), "s0")
    bag = load_bag(str(tmp_path), "s0")
    direct = torch.cat([hipt(regions[i:i + 1].to(DEV)) for i in range(5)]).cpu()
    assert bag.shape == (5, 192) and md(bag, direct.numpy()) < 1e-5 and pt.endswith("s0.pt")
    clam = CLAM_SB(size_arg="hipt_big").eval().to(DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, _ = clam(bag.to(DEV))
    assert logits.shape == (1, 2) and a_raw.shape == (1, 5) and abs(float(y_prob.sum()) - 1) < 1e-5


def test_hipt4k_one_region_from_a_view_off_the_16_byte_grid(hipt):
    """One region per call spreads its patches over the streams and embeds straight from the fp32 pixels with 16-byte loads;
    a region view that starts 4 bytes off a 16-byte boundary must take the converted-copy route, not fail (and give the
    same bits)."""
    W = 2048
    aligned = synth.hash_uniform_torch((1, 3, W, W), 17, device=DEV)
    buf = torch.empty(3 * W * W + 4, dtype=torch.float32, device=DEV)
    off = buf[1:1 + 3 * W * W].view(1, 3, W, W)
    off.copy_(aligned)
    assert aligned.data_ptr() % 16 == 0 and off.data_ptr() % 16 == 4
    hipt.set_compute_dtype("bf16")
    old = hipt.patch_streams
    hipt.patch_streams = 2
    try:
        ref = hipt(aligned)
        got = hipt(off)
    finally:
        hipt.patch_streams = old
        hipt.set_compute_dtype("fp32")
    assert torch.equal(ref, got)


def test_extract_slide_gathered_calls_write_the_same_bits(hipt, tmp_path):
    """Batch-1 loader batches gathered into one HIPT_4K call (feature_store.extract_slide) give bit for bit the features of
    the one-by-one loop, in fp32 and in bf16: a region's rows do not meet another region's anywhere on the path.  (Regions of
    16 patches: like a 4096 x 4096 region's 256, a multiple of 16, so that a call of one region and a call of four take the
    same kernels -- the activation-image path needs whole 16-row fragments; other sizes agree to the bf16 bar only.)"""
    from hipt_abmil_atec23_amd.feature_store import extract_slide
    g = torch.Generator().manual_seed(11)
    regions = torch.randint(0, 256, (5, 1024, 1024, 3), dtype=torch.uint8, generator=g).to(DEV)
    batches = [(regions[i:i + 1], torch.tensor([[4096 * i, 0]])) for i in range(5)]
    for dt in ("fp32", "bf16"):
        hipt.set_compute_dtype(dt)
        try:
            a = torch.load(extract_slide(hipt, batches, str(tmp_path), f"one_{dt}", coalesce=1))
            b = torch.load(extract_slide(hipt, batches, str(tmp_path), f"all_{dt}", coalesce=4))
        finally:
            hipt.set_compute_dtype("fp32")
        assert a.shape == (5, 192) and torch.equal(a, b), dt


def test_extract_slide_4096_regions_gathered_ragged_tail_same_bits(hipt, tmp_path):
    """The same property at the size that matters: 4096 x 4096 regions, whose second-level ViT has 257 token rows per
    region -- one region alone is a small call (<= 1 088 rows), eight gathered ones were not, and a ragged tail of three was
    again.  hipt_vit4k_forward now walks the regions of a call in groups that are small calls, so the kernels a region meets
    do not depend on its company: coalesce = 1, coalesce = 8 (8 + a tail of 3) and coalesce = 5 (5 + 5 + 1) write the same
    features bit for bit, and the coordinates stay aligned.  bf16 = the bench configuration."""
    from hipt_abmil_atec23_amd.feature_store import extract_slide, load_coords
    regions = [synth.hash_uniform_torch((1, 3, 4096, 4096), 900 + i, device=DEV) for i in range(11)]
    batches = [(r, torch.tensor([[4096 * i, 4096 * (i % 3)]], dtype=torch.int64)) for i, r in enumerate(regions)]
    hipt.set_compute_dtype("bf16")
    old_streams = hipt.streams
    try:
        hipt.streams = 2  # (the class default; one region per call runs on one stream)
        one = torch.load(extract_slide(hipt, batches, str(tmp_path), "co1", coalesce=1))
        # one stream: a gathered call is 2 048 patches in ONE pass (its [CLS]-row GEMMs are sliced into small-M launches: capi.hip
        # rows_linear); two streams: two groups of four regions
        for streams, co in ((1, 8), (2, 8), (1, 5)):
            hipt.streams = streams
            got = torch.load(extract_slide(hipt, batches, str(tmp_path), f"co{co}", coalesce=co))
            assert got.shape == (11, 192) and torch.equal(one, got), f"coalesce={co}, {streams} stream(s): max diff {float((one - got).abs().max())}"
            assert np.array_equal(load_coords(str(tmp_path), f"co{co}"), load_coords(str(tmp_path), "co1"))
    finally:
        hipt.streams = old_streams
        hipt.set_compute_dtype("fp32")
    assert load_coords(str(tmp_path), "co1").tolist() == [[4096 * i, 4096 * (i % 3)] for i in range(11)]


def test_region_attention_scores_vs_reference_golden(hipt):
    """HIPT_4K._get_region_attention_scores (hipt_4k.py:121-164), the consumer SURVEY.md 8 f-4 names: attention_256
    [n, 6, 256/s, 256/s] and attention_4k [6, W/s, H/s] of a 1024 x 768 region at scale 4 against the maps of the REFERENCE's
    own ViTs (tests/golden/make_golden.py re-issues hipt_4k.py:135-160 around them): 1e-4 in fp32.  Both maps come from the
    one-query kernels; the [n, 6, 257, 257] tensor is never built."""
    g = golden("hipt4k_attn_1024x768_s4")
    x = synth.hash_uniform_torch((1, 3, 1024, 768), 3, device=DEV)
    hipt.set_compute_dtype("fp32")
    before = N.calls
    patches, a256, a4k = hipt._get_region_attention_scores(x, scale=4)
    assert N.calls > before
    assert isinstance(a256, np.ndarray) and a256.shape == (12, 6, 64, 64) and a4k.shape == (6, 256, 192)
    assert patches.shape == (12, 64, 64, 3) and patches.dtype == np.uint8 and np.array_equal(patches, g["patches_u8"])
    e256, e4k = md(a256, g["attention_256"]), md(a4k, g["attention_4k"])
    print(f"region attention scores vs reference: attention_256 {e256:.1e}, attention_4k {e4k:.1e}")
    assert e256 < TOL and e4k < TOL
    # a raw uint8 image (what the reference passes: a PIL region) goes through eval_transforms and gives the same shapes
    img = ((x[0].permute(1, 2, 0) * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
    p2, b256, b4k = hipt._get_region_attention_scores(img, scale=1)
    assert p2.shape == (12, 256, 256, 3) and b256.shape == (12, 6, 256, 256) and b4k.shape == (6, 1024, 768)
    from hipt_abmil_atec23_amd.hipt_model_utils import eval_transforms, tensorbatch2im
    want = tensorbatch2im(eval_transforms()(img).reshape(3, 4, 256, 3, 256).permute(1, 3, 0, 2, 4).reshape(12, 3, 256, 256))
    assert np.array_equal(p2, want)  # patch k = p1 * h_256 + p2 (index work: bit-exact)
    assert abs(float(b256[:, :, ::16, ::16].sum(axis=(2, 3)).max()) - 1.0) < 0.05  # probabilities minus the [CLS] column


def test_hipt4k_full_region_fp32_and_bf16(hipt):
    """BASELINE config 3 shape: one 4096x4096 region = 256 patches -> ViT-4K over the 16x16 grid."""
    g = golden("hipt4k_4096")
    x = synth.hash_uniform_torch((1, 3, 4096, 4096), 3, device=DEV)
    hipt.set_compute_dtype("fp32")
    d = hipt.forward_asset_dict(x)
    assert md(d["features_cls256"], g["cls256"]) < TOL
    assert md(d["features_cls4k"], g["out"]) < TOL
    hipt.set_compute_dtype("bf16")
    try:
        d = hipt.forward_asset_dict(x)
        c = torch.from_numpy(d["features_cls256"])
        o = torch.from_numpy(d["features_cls4k"])
        print(f"HIPT_4K 4096x4096 bf16 vs reference: cls256 rel-L2 {rel_l2(c, g['cls256']):.2e}, [1,192] rel-L2 {rel_l2(o, g['out']):.2e}, cosine {cosine(o, g['out']):.6f}")
        assert rel_l2(c, g["cls256"]) < 1.3e-2 and cosine(c, g["cls256"]) > 0.9999  # 2 x measured (6.3e-3)
        assert rel_l2(o, g["out"]) < 1.2e-2 and cosine(o, g["out"]) > 0.9999       # 2 x measured (6.0e-3); SURVEY.md 8d bar: 2e-2
        # idempotence: same input, same bits
        assert np.array_equal(hipt.forward_asset_dict(x)["features_cls4k"], d["features_cls4k"])
    finally:
        hipt.set_compute_dtype("fp32")


# ---------------------------------------------------------------------------------------------
# CLAM_SB / Attn_Net_Gated
# ---------------------------------------------------------------------------------------------
CLAM_CASES = [
    ("clam_384_n2000", "hipt_384", (384, 128, 64), 384, (2000, 384), 1, 1, 8, False, 0.0),
    ("clam_384_n777", "hipt_384", (384, 128, 64), 384, (777, 384), 11, 0, 8, False, 0.0),
    ("clam_384_n1", [384, 128, 64], (384, 128, 64), 384, (1, 384), 12, None, 8, False, 0.0),
    ("clam_hipt_big_n500", "hipt_big", (192, 128, 64), 192, (500, 192), 5, 1, 8, False, 0.0),
    ("clam_hipt_smallest_n100", "hipt_smallest", (192, 8, 4), 8, (100, 192), 6, 0, 4, True, 0.0),
    ("clam_small_dropout_n300", "small", (1024, 512, 256), 1024, (300, 1024), 7, None, 8, False, 0.25),
]


def make_clam(size_arg, size, base, k, subtyping, dropout):
    from hipt_abmil_atec23_amd import CLAM_SB
    m = CLAM_SB(gate=True, size_arg=size_arg, dropout=dropout, k_sample=k, n_classes=2, subtyping=subtyping)
    m.load_state_dict(synth.make_state_dict(synth.clam_param_specs(size, dropout=dropout > 0), base), strict=True)
    m.relocate()
    return m.eval()


@pytest.mark.parametrize("name,size_arg,size,base,shape,seed,label,k,subtyping,dropout", CLAM_CASES)
def test_clam_sb_fp32_vs_reference_golden(name, size_arg, size, base, shape, seed, label, k, subtyping, dropout):
    g = golden(name)
    m = make_clam(size_arg, size, base, k, subtyping, dropout)
    h = synth.hash_uniform_torch(shape, seed, device=DEV)
    before = N.calls
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
        att = m(h, attention_only=True)
    assert N.calls >= before + 2  # the HIP entry point ran (no PyTorch path)
    assert a_raw.shape == (1, shape[0]) and md(a_raw, g["A_raw"]) < TOL
    assert md(att, g["attention_only"]) < TOL
    assert md(logits, g["logits"]) < TOL and md(y_prob, g["Y_prob"]) < TOL and md(res["features"], g["M"]) < TOL
    assert y_hat.dtype == torch.int64 and np.array_equal(y_hat.cpu().numpy(), g["Y_hat"])  # bit-exact
    if label is not None:
        with torch.no_grad():
            _, _, _, a2, r = m(h, label=torch.tensor([label], device=DEV), instance_eval=True)
        A = torch.softmax(a2, dim=1)
        assert np.array_equal(torch.topk(A, k)[1][-1].cpu().numpy(), g["top_p"])  # indices bit-exact
        assert np.array_equal(torch.topk(-A, k, dim=1)[1][-1].cpu().numpy(), g["top_n"])
        assert abs(float(r["instance_loss"]) - float(g["instance_loss"])) < TOL
        assert np.array_equal(r["inst_preds"], g["inst_preds"]) and np.array_equal(r["inst_labels"], g["inst_labels"])


def test_clam_sb_bf16_config1():
    g = golden("clam_384_n2000")
    m = make_clam("hipt_384", (384, 128, 64), 384, 8, False, 0.0).set_compute_dtype("bf16")
    h = synth.hash_uniform_torch((2000, 384), 1, device=DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
    # bf16 operands: A_raw spans [-6.3, 3.3]; bars = 2 x measured: 4e-2 abs on A_raw (1.7e-2), 2e-3 rel-L2 on M (5e-4), same argmax
    print(f"CLAM_SB bf16 2000x384 vs reference: A_raw max abs {md(a_raw, g['A_raw']):.2e}, M rel-L2 {rel_l2(res['features'], g['M']):.2e}, logits max abs {md(logits, g['logits']):.2e}")
    assert md(a_raw, g["A_raw"]) < 4e-2
    assert rel_l2(res["features"], g["M"]) < 2e-3 and md(logits, g["logits"]) < 1e-3
    assert np.array_equal(y_hat.cpu().numpy(), g["Y_hat"])


def test_attn_net_gated_vs_reference_golden():
    from hipt_abmil_atec23_amd import Attn_Net_Gated
    g = golden("attn_net_gated_384_256")
    spec = {"attention_a.0.weight": ((256, 384), 0.09, 0.0), "attention_a.0.bias": ((256,), 0.02, 0.0),
            "attention_b.0.weight": ((256, 384), 0.09, 0.0), "attention_b.0.bias": ((256,), 0.02, 0.0),
            "attention_c.weight": ((1, 256), 0.3, 0.0), "attention_c.bias": ((1,), 0.02, 0.0)}
    m = Attn_Net_Gated(L=384, D=256, dropout=0.0, n_classes=1)
    m.load_state_dict(synth.make_state_dict(spec, 9))
    m = m.eval().to(DEV)
    h = synth.hash_uniform_torch((2000, 384), 1, device=DEV)
    with torch.no_grad():
        A, x = m(h)
    assert x is h and A.shape == (2000, 1) and md(A, g["A"]) < TOL
    with torch.no_grad():  # nn.Linear semantics: any leading dims; a wrong feature width is an error, not an out-of-bounds read
        A3, x3 = m(h.view(4, 500, 384))
        assert A3.shape == (4, 500, 1) and torch.equal(A3.view(2000, 1), A) and x3.shape == (4, 500, 384)
        with pytest.raises(RuntimeError, match="does not end in"):
            m(h[:, :256])
        cpu_module = Attn_Net_Gated(L=384, D=256, dropout=0.0, n_classes=1).eval()  # parameters never moved: clean error, no GPU fault
        with pytest.raises(RuntimeError, match="expected all tensors on"):
            cpu_module(h)


@pytest.mark.parametrize("dtype,tolA,tolM", [("fp32", 1e-4, 1e-4), ("bf16", 4e-2, 2e-3)])  # bf16: 2 x measured (2.0e-2 / 3.7e-4)
def test_clam_full_size_100k_properties(dtype, tolA, tolM):
    """BASELINE config 4 shape (100 000 x 384): oracle on the full bag + size-independent properties."""
    n = 100_000
    m = make_clam("hipt_384", (384, 128, 64), 384, 8, False, 0.0).set_compute_dtype(dtype)
    p = synth.make_params_np(synth.clam_param_specs((384, 128, 64)), 384)
    h = synth.hash_uniform_torch((n, 384), 4, device=DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
    r = O.clam_sb_forward(h.cpu().numpy().astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()})
    print(f"CLAM_SB {dtype} 100000x384 vs fp64 oracle: A_raw max abs {md(a_raw, r['A_raw']):.2e}, M rel-L2 {rel_l2(res['features'], r['M']):.2e}, "
          f"logits max abs {md(logits, r['logits']):.2e}")
    assert md(a_raw, r["A_raw"]) < tolA
    assert rel_l2(res["features"], r["M"]) < tolM and md(logits, r["logits"]) < max(tolM, 1e-4)
    assert np.array_equal(y_hat.cpu().numpy(), r["Y_hat"])
    assert abs(float(y_prob.sum()) - 1.0) < 1e-6
    # rows are independent: a 2000-row prefix bag reproduces the same attention logits bit for bit
    with torch.no_grad():
        a_sub = m(h[:2000].contiguous(), attention_only=True)
    assert torch.equal(a_sub[0], a_raw[0, :2000])
    # pooling is permutation invariant
    perm = torch.randperm(n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    with torch.no_grad():
        l2, _, _, a_p, res2 = m(h[perm].contiguous(), return_features=True)
    assert torch.equal(a_p[0], a_raw[0, perm])
    assert md(l2, logits.cpu().numpy()) < 1e-4 and md(res2["features"], res["features"].cpu().numpy()) < 1e-4


def test_clam_two_streams_do_not_share_partials_or_ticket():
    """Every stream gets its own scratch (partials + the self-resetting finish ticket): concurrent CLAM_SB calls on two
    streams, repeated, give the bits of the serial calls; the ticket block is zero again after every call."""
    m = make_clam("hipt_384", (384, 128, 64), 384, 8, False, 0.0).set_compute_dtype("bf16")
    bags = [synth.hash_uniform_torch((50000 + 777 * i, 384), 70 + i, device=DEV).bfloat16() for i in range(2)]
    with torch.no_grad():
        ref = [m(b)[0].clone() for b in bags]
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(20):
            with torch.cuda.stream(s1):
                o1 = m(bags[0])[0]
            with torch.cuda.stream(s2):
                o2 = m(bags[1])[0]
            torch.cuda.synchronize()
            assert torch.equal(o1, ref[0]) and torch.equal(o2, ref[1])
    for key, ws in Fn._workspaces.items():
        if isinstance(key[2], tuple) and key[2][0] == "clam":
            w = m._pack(torch.device(DEV))
            off = N.lib().hipt_clam_ticket_offset(__import__("ctypes").byref(w), 50000)
            assert int(ws[off:off + 256].sum()) == 0, key


def test_clam_generic_then_streaming_model_share_one_workspace():
    """One scratch per (device, stream) serves every CLAM module.  A model on the generic path ('small' [1024,512,256] fp32:
    GEMM + gate + pool kernels writing per-block partials and a running maximum) followed by the bf16 [384,128,64] model on
    the SAME stream: the streaming kernel's arrival counter lives in the first 256 bytes of the workspace, which no other path
    writes, so the second model still combines (outputs written, ticket zero again) -- both orders, repeated."""
    big = make_clam("small", (1024, 512, 256), 1024, 8, False, 0.0)
    fast = make_clam("hipt_384", (384, 128, 64), 384, 8, False, 0.0).set_compute_dtype("bf16")
    hb = synth.hash_uniform_torch((3000, 1024), 81, device=DEV)
    hf = synth.hash_uniform_torch((5000, 384), 82, device=DEV)
    with torch.no_grad():
        ref_f = [t.clone() for t in fast(hf)[:4]]
        ref_b = [t.clone() for t in big(hb)[:4]]
        for _ in range(3):
            out_b = big(hb)[:4]
            out_f = fast(hf)[:4]
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(out_f, ref_f))
            assert all(torch.equal(a, b) for a, b in zip(out_b, ref_b))
    assert float(ref_f[1].sum()) == pytest.approx(1.0, abs=1e-6)
    for key, ws in Fn._workspaces.items():
        if isinstance(key[2], tuple) and key[2][0] == "clam":
            assert int(ws[:256].sum()) == 0, key


def test_clam_sb_bf16_hipt_big_stream_kernel():
    """The aggregator BASELINE configs[4] runs (CLAM_SB 'hipt_big' [192,128,64] in bf16 = abmil_stream_kernel<3>) against the
    reference golden at N = 500 and against the fp64 oracle on an 8 192 x 192 bag (the size of a slide's feature bag).
    Bars = 2 x measured, as test_clam_sb_bf16_config1."""
    g = golden("clam_hipt_big_n500")
    m = make_clam("hipt_big", (192, 128, 64), 192, 8, False, 0.0).set_compute_dtype("bf16")
    h = synth.hash_uniform_torch((500, 192), 5, device=DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
    print(f"CLAM_SB hipt_big bf16 500x192 vs reference: A_raw max abs {md(a_raw, g['A_raw']):.2e}, M rel-L2 {rel_l2(res['features'], g['M']):.2e}, "
          f"logits max abs {md(logits, g['logits']):.2e}")
    assert md(a_raw, g["A_raw"]) < 4e-2 and rel_l2(res["features"], g["M"]) < 2e-3 and md(logits, g["logits"]) < 1e-3
    assert np.array_equal(y_hat.cpu().numpy(), g["Y_hat"])
    n = 8192
    p = synth.make_params_np(synth.clam_param_specs((192, 128, 64)), 192)
    hb = synth.hash_uniform_torch((n, 192), 15, device=DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(hb, return_features=True)
        a_sub = m(hb[:1000].contiguous(), attention_only=True)
    r = O.clam_sb_forward(hb.cpu().numpy().astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()})
    print(f"CLAM_SB hipt_big bf16 8192x192 vs fp64 oracle: A_raw max abs {md(a_raw, r['A_raw']):.2e}, M rel-L2 {rel_l2(res['features'], r['M']):.2e}, "
          f"logits max abs {md(logits, r['logits']):.2e}")
    assert md(a_raw, r["A_raw"]) < 4e-2 and rel_l2(res["features"], r["M"]) < 2e-3 and md(logits, r["logits"]) < 1e-3
    assert np.array_equal(y_hat.cpu().numpy(), r["Y_hat"]) and abs(float(y_prob.sum()) - 1.0) < 1e-6
    assert torch.equal(a_sub[0], a_raw[0, :1000])  # rows are independent
    # a bag long enough that every wave runs several blocks (three slices per block: the odd-slice-count form of the pipeline)
    n2 = 70001
    hc = synth.hash_uniform_torch((n2, 192), 16, device=DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(hc, return_features=True)
    r = O.clam_sb_forward(hc.cpu().numpy().astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()})
    print(f"CLAM_SB hipt_big bf16 {n2}x192 vs fp64 oracle: A_raw max abs {md(a_raw, r['A_raw']):.2e}, M rel-L2 {rel_l2(res['features'], r['M']):.2e}, "
          f"logits max abs {md(logits, r['logits']):.2e}")
    assert md(a_raw, r["A_raw"]) < 4e-2 and rel_l2(res["features"], r["M"]) < 2e-3 and md(logits, r["logits"]) < 1e-3


@pytest.mark.parametrize("n", [1, 31, 32, 33, 97, 4 * 32 * 7 + 5])
def test_clam_stream_kernel_ragged_bags_many_classes_and_the_bound(n):
    """abmil32_kernel's edges: bags of less than one block, one block +- a row, a bag that leaves waves of the last workgroup
    without a block; 9 classes (the general classifier of the in-kernel merge, > 8); and a module whose logit bound is too large
    for the fixed-shift softmax must take the general kernels and agree with the streaming result of its rescaled twin."""
    from hipt_abmil_atec23_amd import CLAM_SB
    m = CLAM_SB(gate=True, size_arg="hipt_384", dropout=0.0, k_sample=1, n_classes=9).eval().to(DEV).set_compute_dtype("bf16")
    sd = synth.make_state_dict(synth.clam_param_specs((384, 128, 64)), 384)
    own = m.state_dict()
    for k, v in sd.items():
        if k in own and own[k].shape == v.shape:
            own[k].copy_(v.to(DEV))
    h = synth.hash_uniform_torch((n, 384), 70 + n, device=DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
        a_only = m(h, attention_only=True)
    p = {k: v.detach().float().cpu().numpy().astype(np.float64) for k, v in m.state_dict().items()}
    w1, b1 = p["attention_net.0.weight"], p["attention_net.0.bias"]
    wa, ba = p["attention_net.2.attention_a.0.weight"], p["attention_net.2.attention_a.0.bias"]
    wb, bb = p["attention_net.2.attention_b.0.weight"], p["attention_net.2.attention_b.0.bias"]
    wc, bc = p["attention_net.2.attention_c.weight"], p["attention_net.2.attention_c.bias"]
    wcl, bcl = p["classifiers.weight"], p["classifiers.bias"]
    x = h.bfloat16().double().cpu().numpy()
    r16 = lambda t: torch.from_numpy(t).bfloat16().double().numpy()
    h1 = np.maximum(x @ r16(w1).T + b1, 0)
    A = (np.tanh(r16(h1) @ r16(wa).T + ba) * (1 / (1 + np.exp(-(r16(h1) @ r16(wb).T + bb))))) @ wc.T + bc
    pw = np.exp(A[:, 0] - A[:, 0].max())
    M = (pw / pw.sum()) @ h1
    lg = M @ wcl.T + bcl
    assert md(a_raw, A.T) < 4e-2 and rel_l2(res["features"], M[None]) < 3e-3 and md(logits, lg[None]) < 2e-3, n
    assert torch.equal(a_only, a_raw) and int(y_hat) == int(lg.argmax()) and abs(float(y_prob.sum()) - 1) < 1e-6
    # a bound beyond the fixed-shift range: the same module with wc scaled up takes the general kernels
    big = CLAM_SB(gate=True, size_arg="hipt_384", dropout=0.0, k_sample=1, n_classes=9).eval().to(DEV).set_compute_dtype("bf16")
    big.load_state_dict(m.state_dict())
    with torch.no_grad():
        big.attention_net[2].attention_c.weight.mul_(4.0)  # sum |wc| ~ 166
        _, _, _, a_big, _ = big(h)
    assert big._pack(h.device).logit_bound > 60
    assert md(a_big, (A - bc).T * 4 + bc) < 0.2


def test_clam_sb_training_shapes_outside_the_training_kernels_keep_autograd():
    """A gated CLAM_SB the training kernels do not take (9 classes > 8) must still train on the GPU: dropout active and
    gradients flowing through the PyTorch-op sequence, not a silent fall-through to the inference kernel."""
    from hipt_abmil_atec23_amd import CLAM_SB
    m = CLAM_SB(gate=True, size_arg="hipt_big", dropout=0.25, k_sample=4, n_classes=9).to(DEV)
    m.relocate()
    m.train()
    h = synth.hash_uniform_torch((64, 192), 91, device=DEV)
    logits, y_prob, y_hat, a_raw, _ = m(h)
    assert logits.shape == (1, 9) and logits.grad_fn is not None and a_raw.grad_fn is not None
    logits.sum().backward()
    assert m.attention_net[0].weight.grad is not None and float(m.attention_net[0].weight.grad.abs().sum()) > 0
    m.eval()
    with torch.no_grad():
        before = N.calls
        out = m(h)
        assert N.calls > before and out[0].grad_fn is None  # inference: the HIP forward


# ---------------------------------------------------------------------------------------------
# boundary behaviour
# ---------------------------------------------------------------------------------------------
def test_errors_are_loud(vit256):
    from hipt_abmil_atec23_amd import CLAM_SB
    with pytest.raises(RuntimeError, match="HIP device"):
        vit256(torch.zeros(1, 3, 256, 256))
    with pytest.raises(ValueError):
        vit256(torch.zeros(1, 4, 256, 256, device=DEV))
    with pytest.raises(RuntimeError, match="envelope"):
        vit256(torch.zeros(1, 3, 512, 512, device=DEV))  # 1025 tokens: outside the on-chip attention envelope
    m = CLAM_SB(size_arg="hipt_384").eval().to(DEV)
    with torch.no_grad():
        with pytest.raises(ValueError):
            m(torch.zeros(0, 384, device=DEV))
        with pytest.raises(ValueError):
            m(torch.zeros(10, 192, device=DEV))
        with pytest.raises(RuntimeError, match="HIP device"):
            m(torch.zeros(10, 384))
    with pytest.raises(RuntimeError, match="libhipt_abmil"):
        N.call("hipt_layernorm", None, 7, None, None, None, 0, 7, 1, 7, 1e-6, None)  # D % 64 != 0


def test_dropin_install_reference_import_paths():
    import hipt_abmil_atec23_amd as amd
    amd.install()
    try:
        from HIPT_4K.hipt_4k import HIPT_4K  # extract_features_fp.py:15
        from HIPT_4K.hipt_model_utils import eval_transforms  # extract_features_fp.py:16
        from models.model_clam import CLAM_MB, CLAM_SB  # utils/core_utils.py:7
        import HIPT_4K.vision_transformer as vits
        import HIPT_4K.vision_transformer4k as vits4k
        assert HIPT_4K is amd.HIPT_4K and CLAM_SB is amd.CLAM_SB and CLAM_MB is amd.CLAM_MB
        assert vits.__dict__['vit_small'] and vits4k.__dict__['vit4k_xs']  # hipt_model_utils.py:54,91 lookups
        t = eval_transforms()(np.full((4, 5, 3), 255, np.uint8))
        assert t.shape == (3, 4, 5) and float(t.max()) == 1.0
    finally:
        from hipt_abmil_atec23_amd.dropin import uninstall
        uninstall()


# ---------------------------------------------------------------------------------------------
# The OUTLIER weight family (round 6; synth.apply_vit_outliers_np): fixtures from the reference's own modules with LayerNorm gains over
# 0.05 ... 20, residual channels of magnitude 50 ... 100 and one peaky head per block (block 0: pre-softmax logits up to +-120).
# fp32 mode: 1e-4 x max|reference| (the residual stream carries |x| ~ 100: fp32 round-off scales with it; the scale is printed);
# bf16 mode: the relative bars of the benign family.  Finite everywhere.
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def vit256_outlier():
    from hipt_abmil_atec23_amd.vision_transformer import vit_small
    m = vit_small(patch_size=16, num_classes=0)
    m.load_state_dict(synth.make_vit_outlier_state_dict(synth.vit_param_specs("vit256"), 256, 6))
    return m.eval().to(DEV)


@pytest.fixture(scope="module")
def hipt_outlier():
    from hipt_abmil_atec23_amd import HIPT_4K
    m = HIPT_4K(None, None, DEV, DEV)
    m.model256.load_state_dict(synth.make_vit_outlier_state_dict(synth.vit_param_specs("vit256"), 256, 6))
    m.model4k.load_state_dict(synth.make_vit_outlier_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096, 6))
    return m.eval().to(DEV)


def test_vit256_outlier_fp32_vs_reference_golden(vit256_outlier):
    g = golden("vit256_outlier")
    m = vit256_outlier.set_compute_dtype("fp32")
    x = synth.hash_uniform_torch((2, 3, 256, 256), 2, device=DEV)
    scale = float(np.abs(g["blk11_rows"]).max())
    print(f"outlier family: residual stream max |x| = {scale:.1f}, largest pre-softmax logit per block {g['logit_absmax_per_block'].round(1).tolist()}")
    tok = m.prepare_tokens(x)
    assert md(tok[:, ROWS], g["tokens_rows"]) < TOL * scale
    t = tok
    for i, blk in enumerate(m.blocks):
        t = blk(t)
        if i in (0, 5, 11):
            assert md(t[:, ROWS], g[f"blk{i}_rows"]) < TOL * scale, i
    out = m(x)
    assert bool(torch.isfinite(out).all()) and md(out, g["out"]) < TOL * float(np.abs(g["out"]).max())
    assert md(m.get_last_selfattention(x)[:, :, 0], g["attn_cls"]) < TOL


def test_vit256_outlier_bf16_small_call_vs_reference_golden(vit256_outlier):
    """two patches = the small-call kernels (gemm_small / lngemm_small / attention) in bf16"""
    g = golden("vit256_outlier")
    m = vit256_outlier.set_compute_dtype("bf16")
    try:
        x = synth.hash_uniform_torch((2, 3, 256, 256), 2, device=DEV)
        out = m(x)
        attn = m.get_last_selfattention(x)
    finally:
        m.set_compute_dtype("fp32")
    print(f"ViT-256 OUTLIER bf16 (small call) vs reference: rel-L2 {rel_l2(out, g['out']):.2e}, cosine {cosine(out, g['out']):.6f}, "
          f"[CLS] attention row max abs {md(attn[:, :, 0], g['attn_cls']):.2e}")
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(attn).all())
    assert rel_l2(out, g["out"]) < 2e-3 and cosine(out, g["out"]) > 0.9999  # 2 x measured (9.3e-4); SURVEY 8d allows 2e-2
    assert md(attn[:, :, 0], g["attn_cls"]) < 2.5e-3  # 2 x measured (1.1e-3)


def test_hipt4k_outlier_fp32_and_bf16_streaming_kernels(hipt_outlier, monkeypatch):
    """1024 x 768 region = 12 patches: in bf16 the whole streaming path (pixel-reading embedding with LayerNorm-1, fused QKV + attention with
    logits of +-120 in block 1's peaky head, proj folded into the fused MLP with the clamped 3-coefficient GELU, [CLS]-pruned last block)
    and ViT-4K's small-call kernels, on the outlier family; also with the attention as two kernels (HIPT_NO_FUSED_ATTN) within the same bar."""
    g = golden("hipt4k_outlier_1024")
    h = hipt_outlier
    x = synth.hash_uniform_torch((1, 3, 1024, 768), 3, device=DEV)
    h.set_compute_dtype("fp32")
    d = h.forward_asset_dict(x)
    s256, s4k = float(np.abs(g["cls256"]).max()), float(np.abs(g["out"]).max())
    print(f"HIPT_4K OUTLIER fp32: cls256 max abs err {md(d['features_cls256'], g['cls256']):.2e} (scale {s256:.1f}), out {md(d['features_cls4k'], g['out']):.2e} (scale {s4k:.1f})")
    assert md(d["features_cls256"], g["cls256"]) < TOL * s256 and md(d["features_cls4k"], g["out"]) < TOL * s4k
    h.set_compute_dtype("bf16")
    try:
        before = N.calls
        d = h.forward_asset_dict(x)
        assert N.calls > before
        monkeypatch.setenv("HIPT_NO_FUSED_ATTN", "1")
        d2 = h.forward_asset_dict(x)
    finally:
        monkeypatch.delenv("HIPT_NO_FUSED_ATTN", raising=False)
        h.set_compute_dtype("fp32")
    for name, dd in (("fused attention", d), ("two-kernel attention", d2)):
        c, o = torch.from_numpy(dd["features_cls256"]), torch.from_numpy(dd["features_cls4k"])
        print(f"HIPT_4K OUTLIER bf16, {name}: cls256 rel-L2 {rel_l2(c, g['cls256']):.2e} cosine {cosine(c, g['cls256']):.6f}; "
              f"[1,192] rel-L2 {rel_l2(o, g['out']):.2e} cosine {cosine(o, g['out']):.6f}")
        assert bool(torch.isfinite(c).all()) and bool(torch.isfinite(o).all()), name
        assert rel_l2(c, g["cls256"]) < 2e-3 and cosine(c, g["cls256"]) > 0.9999, name  # 2 x measured (9.5e-4); SURVEY 8d allows 2e-2
        assert rel_l2(o, g["out"]) < 1e-3 and cosine(o, g["out"]) > 0.9999, name       # 2 x measured (3.4e-4)


@pytest.mark.parametrize("blk", [0, 3])  # block 0: logits up to +-120 in head 0; block 3: +-40 in head 3
def test_attention_unit_outlier_weights_large_logits(vit256_outlier, blk):
    """hipt_vit_attention_unit with the outlier family's block weights on operands with LayerNorm-output statistics of that family (a few
    channels of magnitude ~30): the fused bf16 kernel's running maximum / the two-kernel form against fp32 torch on the same bf16 operands."""
    m = vit256_outlier.set_compute_dtype("bf16")
    try:
        nseq = 48
        pk = m._tokens(synth.hash_uniform_torch((1, 3, 256, 256), 2, device=DEV))[0]
        M = nseq * 257
        x = synth.hash_uniform_torch((M, 384), 91 + blk, device=DEV) * 1.7
        big = torch.from_numpy(synth.hash_u32_np(384, 5).astype(np.int64) % 50 == 0).to(DEV)
        x[:, big] *= 12.0
        x = x.bfloat16()
        xi = _to_image(x)
        ws = Fn.workspace(torch.device(DEV), N.lib().hipt_vit_workspace_bytes(pk.ref, nseq))
        outs = []
        for fused in (1, 0):
            o = torch.full((M, 384), float("nan"), dtype=torch.bfloat16, device=DEV)
            N.call("hipt_vit_attention_unit", pk.ref, blk, N.ptr(xi), nseq, N.ptr(o), fused, N.ptr(ws), ws.numel(), N.stream_ptr(torch.device(DEV)))
            torch.cuda.synchronize()
            outs.append(_from_image(o).float())
        att = m.blocks[blk].attn
        wq = att.qkv.weight.detach().bfloat16().float()
        qkv = (x.float() @ wq.t() + att.qkv.bias.detach().float()).bfloat16().float().view(nseq, 257, 3, 6, 64).permute(2, 0, 3, 1, 4)
        lg = (qkv[0] @ qkv[1].transpose(-1, -2)) * att.scale
        ref = (torch.softmax(lg, dim=-1) @ qkv[2]).transpose(1, 2).reshape(M, 384)
    finally:
        m.set_compute_dtype("fp32")
    rmax = float(ref.abs().max())
    print(f"attention unit, outlier block {blk}: largest |logit| {float(lg.abs().max()):.1f}, |ref| max {rmax:.2f}")
    assert float(lg.abs().max()) > 30
    for name, o in zip(("fused", "two kernels"), outs):
        assert bool(torch.isfinite(o).all()), name
        e = float((o - ref).abs().max())
        print(f"   {name}: max |err| vs fp32 torch {e:.2e}")
        assert e < 3e-2 * rmax, name  # 2 x measured (block 3, two kernels: 1.5e-2: logits of +-330 flip near-ties under bf16 q, k)


@pytest.mark.parametrize("tag,bound", [("lo", 50.0), ("hi", 70.0)])
def test_clam_outlier_bounds_both_sides_of_the_fixed_shift(tag, bound):
    """CLAM_SB [384,128,64] with attention_c rescaled so that sum |wc| = 50 (bf16: the one-launch fixed-shift softmax kernel, bound < 60) and
    70 (the general kernels), against the reference's own outputs; fp32 at 1e-4."""
    from hipt_abmil_atec23_amd import CLAM_SB
    g = golden("clam_outlier_n2000")
    p = synth.scale_clam_attention_c_np(synth.make_params_np(synth.clam_param_specs((384, 128, 64)), 384), bound)
    m = CLAM_SB(gate=True, size_arg="hipt_384", dropout=0.0, k_sample=8, n_classes=2)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=True)
    m.relocate()
    m = m.eval()
    h = synth.hash_uniform_torch((2000, 384), 1, device=DEV)
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
    assert md(a_raw, g[f"{tag}_A_raw"]) < TOL and md(res["features"], g[f"{tag}_M"]) < TOL and md(logits, g[f"{tag}_logits"]) < TOL
    assert np.array_equal(y_hat.cpu().numpy(), g[f"{tag}_Y_hat"])
    m.set_compute_dtype("bf16")
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, res = m(h, return_features=True)
    pk = m._pack(h.device)
    assert (pk.logit_bound < 60) == (tag == "lo")
    span = float(np.abs(g[f"{tag}_A_raw"]).max())
    print(f"CLAM_SB outlier '{tag}' (sum|wc| = {pk.logit_bound:.1f}) bf16 vs reference: A_raw max abs {md(a_raw, g[f'{tag}_A_raw']):.2e} (|A_raw| max {span:.1f}), "
          f"M rel-L2 {rel_l2(res['features'], g[f'{tag}_M']):.2e}, logits max abs {md(logits, g[f'{tag}_logits']):.2e}")
    assert bool(torch.isfinite(a_raw).all()) and bool(torch.isfinite(logits).all())
    assert md(a_raw, g[f"{tag}_A_raw"]) < 1e-2 * max(span, 4.0)   # bf16 operands: the benign family's 4e-2 at |A_raw| ~ 6, scaled with the logits
    assert rel_l2(res["features"], g[f"{tag}_M"]) < 2e-2 and md(logits, g[f"{tag}_logits"]) < 2e-2
    assert np.array_equal(y_hat.cpu().numpy(), g[f"{tag}_Y_hat"])


def test_extract_slide_host_batches_pipelined_same_bits(hipt, tmp_path):
    """The loop WITH its H2D hop (extract_features_fp.py:162-166; feature_store._HostFeed): loader batches in host memory -- pinned (a
    DataLoader with pin_memory) and pageable, uint8 interleaved and fp32 -- are copied on a copy stream into double-buffered gather buffers
    while the previous call computes, features read back one call late.  The files are the resident loop's, bit for bit (ragged tail, more
    calls than buffers, coords aligned)."""
    from hipt_abmil_atec23_amd.feature_store import extract_slide, load_coords
    hipt.set_compute_dtype("bf16")
    try:
        n = 11  # coalesce 4: calls of 4, 4, 3 -> both gather buffers re-used
        f32 = [synth.hash_uniform_torch((1, 3, 1024, 1024), 300 + i, device=DEV) for i in range(n)]
        u8 = [((r * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous() for r in f32]
        coords = [torch.tensor([[1024 * i, 7 * i]], dtype=torch.int64) for i in range(n)]
        for kind, regs in (("f32", f32), ("u8", u8)):
            ref = torch.load(extract_slide(hipt, list(zip(regs, coords)), str(tmp_path), f"res_{kind}", coalesce=4))
            for mem in ("pinned", "pageable"):
                host = [r.cpu().pin_memory() if mem == "pinned" else r.cpu() for r in regs]
                before = N.calls
                got = torch.load(extract_slide(hipt, list(zip(host, coords)), str(tmp_path), f"{mem}_{kind}", coalesce=4))
                assert N.calls > before
                assert torch.equal(got, ref), (kind, mem)
                assert np.array_equal(load_coords(str(tmp_path), f"{mem}_{kind}"), torch.cat(coords).numpy())
            one = torch.load(extract_slide(hipt, list(zip([r.cpu() for r in regs], coords)), str(tmp_path), f"one_{kind}", coalesce=1))
            assert torch.equal(one, ref), kind
        # shapes that are not gathered (6 patches: no whole 16-row fragments) go one loader batch per call, from host memory too; and a slide
        # whose batches change shape mid-way (a staged gather is closed, the new shape starts its own)
        small = [synth.hash_uniform_torch((1, 3, 512, 768), 340 + i, device=DEV) for i in range(3)]
        mixed = small[:2] + f32[:3] + small[2:]
        mc = [torch.tensor([[i, i]], dtype=torch.int32) for i in range(len(mixed))]
        ref = torch.load(extract_slide(hipt, list(zip(mixed, mc)), str(tmp_path), "mixed_res", coalesce=4))
        got = torch.load(extract_slide(hipt, list(zip([r.cpu().pin_memory() for r in mixed], mc)), str(tmp_path), "mixed_host", coalesce=4))
        assert got.shape == (6, 192) and torch.equal(got, ref)
        assert load_coords(str(tmp_path), "mixed_host").dtype == np.int32
    finally:
        hipt.set_compute_dtype("fp32")


def test_step_bits_do_not_depend_on_stream_concurrency_or_the_run(hipt):
    """Run-to-run determinism under stream concurrency (round 6: the uint8 embedding's miscounted ring wait showed ONLY as run-to-run differences
    when two streams embedded at once -- tools/dbg_u8.py, tools/soak_step_determinism.py): six 4096 x 4096 regions over 3 / 2 / 1 streams, fp32 and
    interleaved uint8 input, repeated: every repeat and every stream count returns the same bits."""
    hipt.set_compute_dtype("bf16")
    old = hipt.streams
    try:
        x = synth.hash_uniform_torch((6, 3, 4096, 4096), 3, device=DEV)
        x8 = ((x * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        for name, inp in (("fp32", x), ("uint8", x8)):
            ref = None
            for streams in (3, 2, 1):
                hipt.streams = streams
                for rep in range(4):
                    o = hipt(inp)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = o.clone()
                    assert torch.equal(o, ref), (name, streams, rep)
    finally:
        hipt.streams = old
        hipt.set_compute_dtype("fp32")
