#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Run ONLY in the build container (needs the read-only reference checkout):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference]

What is imported from the reference (unchanged, from where it lies):
  * HIPT_4K.vision_transformer      (VisionTransformer, vit_small)
  * HIPT_4K.vision_transformer4k    (VisionTransformer4K, vit4k_xs)
  * models.model_clam               (CLAM_SB, CLAM_MB, Attn_Net_Gated)
``torchvision`` is not installed here and is never used on the arithmetic path
(vision_transformer4k.py:17-18 import it without using it; model_clam.py:4 pulls it in via
utils/utils.py:10), so an empty in-process module object satisfies those import
statements.  HIPT_4K/hipt_4k.py itself cannot be imported (TabError in
hipt_model_utils.py:72,109 plus h5py/cv2/openslide imports), so the dozen lines of glue in
HIPT_4K.forward (hipt_4k.py:63-75) are re-issued here with the same torch/einops calls
around the reference's real ViT modules.

Inputs and weights come from hipt_abmil_atec23_amd.synth (integer hash, reproducible
anywhere), so only OUTPUTS are stored.  Nothing of the reference's source is written out.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from hipt_abmil_atec23_amd import synth  # noqa: E402


def import_reference(ref):
    sys.dont_write_bytecode = True
    sys.path.insert(0, ref)
    for n in ("torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.models"):
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    tv = sys.modules["torchvision"]
    tv.transforms, tv.datasets, tv.models = (sys.modules["torchvision.transforms"],
                                             sys.modules["torchvision.datasets"],
                                             sys.modules["torchvision.models"])
    import HIPT_4K.vision_transformer as vits
    import HIPT_4K.vision_transformer4k as vits4k
    import models.model_clam as clam
    return vits, vits4k, clam


def load(model, specs, base_seed=0):
    sd = synth.make_state_dict(specs, base_seed)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    # instance_loss_fn etc. carry no tensors; everything else must match exactly
    assert not unexpected and not [m for m in missing if "instance_loss_fn" not in m], (missing, unexpected)
    return model.eval()


SKIP_EXISTING = False


def save(name, **arrs):
    if SKIP_EXISTING and os.path.isfile(os.path.join(HERE, name + ".npz")):
        print(f"kept {name}.npz")
        return
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: " + ", ".join(f"{k}{list(v.shape)}" for k, v in out.items()))


ROWS = [0, 1, 128, 256]


def clam_train_goldens(clam):
    """Training steps of the reference's CLAM modules with autograd: the loss of train_loop_clam
    (utils/core_utils.py:300-348: bag_weight * CE(logits, label) + (1 - bag_weight) * instance_loss, bag_weight 0.7) or of
    train_loop (:373-426: CE alone), backward(), and the gradient of every parameter."""
    def step(model, h, label, instance_eval, bag_weight=0.7):
        model.train()
        model.zero_grad()
        lab = torch.tensor([label])
        logits, y_prob, y_hat, a_raw, res = model(h, label=lab, instance_eval=instance_eval, return_features=True)
        loss = torch.nn.functional.cross_entropy(logits, lab)
        total = bag_weight * loss + (1 - bag_weight) * res["instance_loss"] if instance_eval else loss
        total.backward()
        d = dict(logits=logits, Y_prob=y_prob, Y_hat=y_hat, A_raw=a_raw, M=res["features"], loss=total)
        if instance_eval:
            d["instance_loss"] = res["instance_loss"]
            d["inst_preds"], d["inst_labels"] = res["inst_preds"], res["inst_labels"]
        for k, p in model.named_parameters():
            d["grad." + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        return d

    with torch.enable_grad():
        for n, seed in ((15, 21), (100, 22), (2000, 23)):
            m = load(clam.CLAM_SB(size_arg="hipt_big", k_sample=8), synth.clam_param_specs((192, 128, 64)), 192)
            h = synth.hash_uniform_torch((n, 192), seed=seed)
            save(f"clam_grad_hipt_big_n{n}", **step(m, h, 1, True))
        m = load(clam.CLAM_SB(size_arg="hipt_big", k_sample=8), synth.clam_param_specs((192, 128, 64)), 192)
        save("clam_grad_hipt_big_n100_bagonly", **step(m, synth.hash_uniform_torch((100, 192), seed=22), 0, False))
        m = load(clam.CLAM_SB(size_arg="hipt_smallest", k_sample=4, subtyping=True), synth.clam_param_specs((192, 8, 4)), 8)
        save("clam_grad_hipt_smallest_n100", **step(m, synth.hash_uniform_torch((100, 192), seed=6), 0, True))
        mb = load(clam.CLAM_MB(size_arg="hipt_big", k_sample=8, n_classes=3, subtyping=True),
                  synth.clam_param_specs((192, 128, 64), n_classes=3, multi=True), 193)
        save("clam_mb_grad_hipt_big_n100", **step(mb, synth.hash_uniform_torch((100, 192), seed=24), 2, True))
    with torch.no_grad():  # CLAM_MB eval forward (no instance branch)
        mb.eval()
        logits, y_prob, y_hat, a_raw, res = mb(synth.hash_uniform_torch((333, 192), seed=25), return_features=True)
        save("clam_mb_hipt_big_n333", logits=logits, Y_prob=y_prob, Y_hat=y_hat, A_raw=a_raw, M=res["features"])


@torch.no_grad()
def outlier_goldens(vits, vits4k, clam):
    """The OUTLIER weight family (hipt_abmil_atec23_amd/synth.py, round 6): LayerNorm gains over 0.05 ... 20, residual
    channels of magnitude 50 ... 100, one peaky head per block (logits up to +-120 in block 0), CLAM attention_c on both sides of the
    fixed-shift softmax's bound of 60 -- through the reference's own modules."""
    from einops import rearrange
    m256 = vits.vit_small(patch_size=16, num_classes=0)
    m256.load_state_dict(synth.make_vit_outlier_state_dict(synth.vit_param_specs("vit256"), 256, 6))
    m256.eval()
    x = synth.hash_uniform_torch((2, 3, 256, 256), seed=2)
    tok = m256.prepare_tokens(x)
    t, taps = tok, {}
    for i, blk in enumerate(m256.blocks):
        t = blk(t)
        if i in (0, 5, 11):
            taps[i] = t
    out = m256.norm(t)[:, 0]
    assert torch.equal(out, m256(x))
    # the largest pre-softmax logit of every block (what the softmax kernels must survive), recorded with the fixture
    lmax, tt = [], tok
    for blk in m256.blocks:
        qkv = blk.attn.qkv(blk.norm1(tt)).reshape(2, 257, 3, 6, 64).permute(2, 0, 3, 1, 4)
        lmax.append(float(((qkv[0] @ qkv[1].transpose(-2, -1)) * blk.attn.scale).abs().max()))
        tt = blk(tt)
    save("vit256_outlier", tokens_rows=tok[:, ROWS], blk0_rows=taps[0][:, ROWS], blk5_rows=taps[5][:, ROWS], blk11_rows=taps[11][:, ROWS],
         out=out, attn_cls=m256.get_last_selfattention(x)[:, :, 0, :], logit_absmax_per_block=np.asarray(lmax, np.float32))

    m4k = vits4k.vit4k_xs(num_classes=0)
    m4k.load_state_dict(synth.make_vit_outlier_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096, 6))
    m4k.eval()
    r1k = synth.hash_uniform_torch((1, 3, 1024, 768), seed=3)
    b = rearrange(r1k.unfold(2, 256, 256).unfold(3, 256, 256), 'b c p1 p2 w h -> (b p1 p2) c w h')  # hipt_4k.py:64-65
    f = m256(b)  # :68-72
    grid = f.reshape(4, 3, 384).transpose(0, 1).transpose(0, 2).unsqueeze(dim=0)  # :73
    save("hipt4k_outlier_1024", out=m4k.forward(grid), cls256=f)  # :75

    d = {}
    for tag, bound in (("lo", 50.0), ("hi", 70.0)):
        c384 = clam.CLAM_SB(gate=True, size_arg="hipt_big", dropout=0.0, k_sample=8, n_classes=2)
        c384.attention_net[0] = torch.nn.Linear(384, 128)
        p = synth.scale_clam_attention_c_np(synth.make_params_np(synth.clam_param_specs((384, 128, 64)), 384), bound)
        c384.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}, strict=False)
        c384.eval()
        h = synth.hash_uniform_torch((2000, 384), seed=1)
        logits, y_prob, y_hat, a_raw, res = c384(h, return_features=True)
        d.update({f"{tag}_logits": logits, f"{tag}_Y_prob": y_prob, f"{tag}_Y_hat": y_hat, f"{tag}_A_raw": a_raw, f"{tag}_M": res["features"],
                  f"{tag}_bound": np.float32(np.abs(p["attention_net.2.attention_c.weight"].astype(np.float64)).sum())})
    save("clam_outlier_n2000", **d)


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="'train': only the CLAM training-step fixtures; 'outlier': only the outlier-family fixtures; 'new': skip fixtures that already exist")
    ap.add_argument("--skip-4096", action="store_true")
    args = ap.parse_args()
    global SKIP_EXISTING
    SKIP_EXISTING = args.only == "new"
    vits, vits4k, clam = import_reference(args.ref)
    if args.only == "outlier":
        outlier_goldens(vits, vits4k, clam)
        return
    clam_train_goldens(clam)
    if args.only == "train":
        return
    outlier_goldens(vits, vits4k, clam)
    torch.manual_seed(0)

    # ---- (1)+(3) ViT-256, full config, two 256x256 patches (BASELINE config 2 uses patch 0) ----
    m256 = load(vits.vit_small(patch_size=16, num_classes=0), synth.vit_param_specs("vit256"), 256)
    x = synth.hash_uniform_torch((2, 3, 256, 256), seed=2)
    tok = m256.prepare_tokens(x)
    t = tok
    taps = {}
    for i, blk in enumerate(m256.blocks):
        t = blk(t)
        if i in (0, 5, 11):
            taps[i] = t
    out = m256.norm(t)[:, 0]
    assert torch.equal(out, m256(x))
    attn = m256.get_last_selfattention(x)
    pos = m256.interpolate_pos_encoding(tok, 256, 256)
    save("vit256_full", tokens_rows=tok[:, ROWS], blk0_rows=taps[0][:, ROWS], blk5_rows=taps[5][:, ROWS],
         blk11_rows=taps[11][:, ROWS], out=out, attn_cls=attn[:, :, 0, :], attn_row200=attn[:, :, 200, :], pos=pos)

    # ---- (2) reduced ViT (depth 2, D 64, 2 heads -> dh 32), non-square 64x96 input: pins pos-interp + ordering ----
    cfg = dict(embed_dim=64, depth=2, num_heads=2)
    ms = vits.VisionTransformer(patch_size=16, num_classes=0, mlp_ratio=4, qkv_bias=True,
                                norm_layer=vits.partial(torch.nn.LayerNorm, eps=1e-6), **cfg)
    load(ms, synth.vit_param_specs("vit256", **cfg), 64)
    xs = synth.hash_uniform_torch((2, 3, 64, 96), seed=22)
    toks = ms.prepare_tokens(xs)
    save("vit_small_cfg", tokens=toks, out=ms(xs), attn=ms.get_last_selfattention(xs),
         inter=torch.stack(ms.get_intermediate_layers(xs, n=2)), pos=ms.interpolate_pos_encoding(toks, 64, 96))

    # ---- (4) ViT-4K, full config: 16x16 grid and a 3x4 grid ----
    m4k = load(vits4k.vit4k_xs(num_classes=0), synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096)
    g16 = synth.hash_uniform_torch((1, 384, 16, 16), seed=4)
    g34 = synth.hash_uniform_torch((2, 384, 3, 4), seed=44)
    tok4 = m4k.prepare_tokens(g16)
    save("vit4k", out16=m4k(g16), out34=m4k(g34), tokens16_rows=tok4[:, ROWS],
         pos16=m4k.interpolate_pos_encoding(tok4, 16, 16),
         attn_cls16=m4k.get_last_selfattention(g16)[:, :, 0, :],
         pos34=m4k.interpolate_pos_encoding(m4k.prepare_tokens(g34), 3, 4))

    # ---- (5) HIPT_4K composite: the glue of hipt_4k.py:63-75 around the reference ViTs ----
    from einops import rearrange

    def hipt_forward(region):
        _, _, w, h = region.shape
        w_256, h_256 = w // 256, h // 256  # prepare_img_tensor is the identity for multiples of 256 (:325-329)
        b = region.unfold(2, 256, 256).unfold(3, 256, 256)  # :64
        b = rearrange(b, 'b c p1 p2 w h -> (b p1 p2) c w h')  # :65
        feats = [m256(b[i:i + 256]) for i in range(0, b.shape[0], 256)]  # :68-70
        f = torch.vstack(feats)  # :72
        grid = f.reshape(w_256, h_256, 384).transpose(0, 1).transpose(0, 2).unsqueeze(dim=0)  # :73
        return m4k.forward(grid), f  # :75

    r1k = synth.hash_uniform_torch((1, 3, 1024, 768), seed=3)
    o, f = hipt_forward(r1k)
    save("hipt4k_1024x768", out=o, cls256=f)
    if not args.skip_4096:
        r4k = synth.hash_uniform_torch((1, 3, 4096, 4096), seed=3)
        o, f = hipt_forward(r4k)
        save("hipt4k_4096", out=o, cls256=f)

    # ---- (5b) HIPT_4K._get_region_attention_scores: hipt_4k.py:135-160 re-issued around the reference ViTs (tensor half only) ----
    def region_attention(x, scale):
        _, _, w, h = x.shape
        w_256, h_256 = w // 256, h // 256
        b = x.unfold(2, 256, 256).unfold(3, 256, 256)  # :138
        b = rearrange(b, 'b c p1 p2 w h -> (b p1 p2) c w h')  # :139
        cls = m256(b)  # :141
        a256 = m256.get_last_selfattention(b)  # :143
        nh = a256.shape[1]
        a256 = a256[:, :, 0, 1:].reshape(w_256 * h_256, nh, -1)  # :145 (the reference writes 256 = its w_256 * h_256)
        a256 = a256.reshape(w_256 * h_256, nh, 16, 16)  # :146
        a256 = torch.nn.functional.interpolate(a256, scale_factor=int(16 / scale), mode="nearest")  # :147
        grid = cls.reshape(w_256, h_256, 384).transpose(0, 1).transpose(0, 2).unsqueeze(dim=0)  # :149
        a4k = m4k.get_last_selfattention(grid)  # :153
        nh = a4k.shape[1]
        a4k = a4k[0, :, 0, 1:].reshape(nh, -1).reshape(nh, w_256, h_256)  # :155-156
        a4k = torch.nn.functional.interpolate(a4k.unsqueeze(0), scale_factor=int(256 / scale), mode="nearest")[0]  # :157
        if scale != 1:
            b = torch.nn.functional.interpolate(b, scale_factor=(1 / scale), mode="nearest")  # :160
        return b, a256, a4k

    b, a256, a4k = region_attention(r1k, 4)
    save("hipt4k_attn_1024x768_s4", attention_256=a256, attention_4k=a4k, patches_u8=((b.permute(0, 2, 3, 1) + 1) / 2.0 * 255.0).numpy().astype(np.uint8))

    # ---- (6)+(7) CLAM_SB / Attn_Net_Gated ----
    def clam_pack(model, h, label=None, k=8):
        logits, y_prob, y_hat, a_raw, res = model(h, return_features=True)
        d = dict(logits=logits, Y_prob=y_prob, Y_hat=y_hat, A_raw=a_raw, M=res["features"],
                 attention_only=model(h, attention_only=True))
        if label is not None:
            A = torch.softmax(a_raw, dim=1)
            d["top_p"] = torch.topk(A, k)[1][-1]
            d["top_n"] = torch.topk(-A, k, dim=1)[1][-1]
            _, _, _, _, r = model(h, label=torch.tensor([label]), instance_eval=True)
            d["instance_loss"] = r["instance_loss"]
            d["inst_preds"] = r["inst_preds"]
            d["inst_labels"] = r["inst_labels"]
        return d

    # config 1: 2000 x 384 bag, widths [384,128,64] (SURVEY.md §8d "384 sizing"): the reference
    # size_dict has no 384-input entry, so the first Linear of 'hipt_big' is swapped for a 384-input one
    c384 = clam.CLAM_SB(gate=True, size_arg="hipt_big", dropout=0.0, k_sample=8, n_classes=2)
    c384.attention_net[0] = torch.nn.Linear(384, 128)
    load(c384, synth.clam_param_specs((384, 128, 64)), 384)
    h = synth.hash_uniform_torch((2000, 384), seed=1)
    save("clam_384_n2000", **clam_pack(c384, h, label=1))
    h = synth.hash_uniform_torch((777, 384), seed=11)  # ragged N (not a multiple of any tile)
    save("clam_384_n777", **clam_pack(c384, h, label=0))
    h = synth.hash_uniform_torch((1, 384), seed=12)  # single-instance bag
    save("clam_384_n1", **clam_pack(c384, h))

    cbig = load(clam.CLAM_SB(size_arg="hipt_big", k_sample=8), synth.clam_param_specs((192, 128, 64)), 192)
    h = synth.hash_uniform_torch((500, 192), seed=5)
    save("clam_hipt_big_n500", **clam_pack(cbig, h, label=1))

    csub = load(clam.CLAM_SB(size_arg="hipt_smallest", k_sample=4, subtyping=True),
                synth.clam_param_specs((192, 8, 4)), 8)
    h = synth.hash_uniform_torch((100, 192), seed=6)
    save("clam_hipt_smallest_n100", **clam_pack(csub, h, label=0, k=4))

    cdrop = load(clam.CLAM_SB(size_arg="small", dropout=0.25),
                 synth.clam_param_specs((1024, 512, 256), dropout=True), 1024)
    h = synth.hash_uniform_torch((300, 1024), seed=7)
    save("clam_small_dropout_n300", **clam_pack(cdrop, h))

    gated = clam.Attn_Net_Gated(L=384, D=256, dropout=0.0, n_classes=1)
    gp = {k: (s, sc, off) for k, (s, sc, off) in {
        "attention_a.0.weight": ((256, 384), 0.09, 0.0), "attention_a.0.bias": ((256,), 0.02, 0.0),
        "attention_b.0.weight": ((256, 384), 0.09, 0.0), "attention_b.0.bias": ((256,), 0.02, 0.0),
        "attention_c.weight": ((1, 256), 0.3, 0.0), "attention_c.bias": ((1,), 0.02, 0.0)}.items()}
    load(gated, gp, 9)
    h = synth.hash_uniform_torch((2000, 384), seed=1)
    A, xr = gated(h)
    assert xr is h
    save("attn_net_gated_384_256", A=A)

    # shipped demo checkpoint: records the key layout the build's CLAM_SB must load with strict=True
    ck = os.path.join(args.ref, "heatmaps/demo/ckpts/s_0_checkpoint.pt")
    if os.path.isfile(ck):
        sd = torch.load(ck, map_location="cpu")
        with open(os.path.join(HERE, "demo_ckpt_keys.txt"), "w") as fh:
            for k, v in sd.items():
                fh.write(f"{k} {list(v.shape)}\n")
        print("wrote demo_ckpt_keys.txt")


if __name__ == "__main__":
    main()
