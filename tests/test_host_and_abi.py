"""CPU tests (no GPU): the C-ABI library loads and exports every declared symbol, the host mirror
keeps the reference's state-dict / constructor / import-path contract, and CPU inputs fail loudly
(there is no CPU execution path in the product)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from hipt_abmil_atec23_amd import _native as N
from hipt_abmil_atec23_amd import synth


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hipt_abmil.h")).read()
    declared = set(re.findall(r"\b(hipt_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    lib = N.lib()  # loads the .so built by __graft_entry__.build(); raises if missing
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    assert lib.hipt_abi_version() == N.ABI_VERSION


def test_struct_layouts_match_header():
    import ctypes as C
    # hipt_vit_weights: 8 x 4-byte scalars, 6 pointers, 1 pointer
    assert C.sizeof(N.VitWeights) == 10 * 4 + 7 * 8
    assert C.sizeof(N.BlockWeights) == 15 * 8 + 8 + 8  # 12 matrices / vectors + 3 optional packed images + the MLP image's format + the fused-attention image
    assert N.BlockWeights.qkv_att_pk.offset == 15 * 8 + 8
    assert C.sizeof(N.ImageLayout) == 4 * 4 + 3 * 8
    assert C.sizeof(N.ClamWeights) == 6 * 4 + 8 * 8 + 8 + 8 and N.ClamWeights.logit_bound.offset == 6 * 4 + 8 * 8
    assert N.VitWeights.ln_eps.offset == 28 and N.VitWeights.attn_scale.offset == 32 and N.VitWeights.embed_w.offset == 40


def test_struct_layouts_match_the_c_header_as_gcc_sees_it(tmp_path):
    """sizeof / offsetof of every ABI struct from include/hipt_abmil.h itself (compiled by gcc) against the ctypes mirror."""
    import ctypes as C
    import subprocess

    structs = {"hipt_clam_weights": N.ClamWeights, "hipt_vit_weights": N.VitWeights, "hipt_block_weights": N.BlockWeights,
               "hipt_image_layout": N.ImageLayout}
    lines = []
    for cname, cls in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hipt_abmil.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_integration_md_binding_is_the_current_abi():
    """INTEGRATION.md section 3 shows the stand-alone ctypes binding a maintainer would copy: its code block must run against the
    built library and declare hipt_clam_weights exactly as the header / the shipped binding do (it was one ABI behind)."""
    import ctypes as C

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 3. Binding the C ABI directly"):]
    block = sec[sec.index("```python") + len("```python"):]
    block = block[:block.index("```")]
    N.lib()  # (torch first, then the library: the load order the document states)
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        exec(compile(block, "INTEGRATION.md#3", "exec"), ns)
    finally:
        os.chdir(cwd)
    doc_cls = ns["ClamWeights"]
    assert C.sizeof(doc_cls) == C.sizeof(N.ClamWeights)
    assert [(n, getattr(doc_cls, n).offset, getattr(doc_cls, n).size) for n, _ in doc_cls._fields_] == \
           [(n, getattr(N.ClamWeights, n).offset, getattr(N.ClamWeights, n).size) for n, _ in N.ClamWeights._fields_]
    assert ns["lib"].hipt_abi_version() == N.ABI_VERSION
    assert callable(ns["clam_sb_forward"])


def test_integration_md_names_only_switches_the_code_reads():
    """INTEGRATION.md section 6 lists the environment switches: each one must be read somewhere in the package (C or Python), and
    every switch the library reads (csrc/common.h's hipt_env_on calls) must be listed -- the table was two rounds stale once."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 6. Environment switches"):]
    listed = set(re.findall(r"`(HIPT_[A-Z0-9_]+)(?:=[^`]*)?`", sec.split("\n## ")[0]))
    src = ""
    pkg = os.path.join(ROOT, "hipt_abmil_atec23_amd")
    for dirpath, _, files in os.walk(pkg):
        if os.sep + "build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src += open(os.path.join(dirpath, f)).read()
    read_by_lib = set(re.findall(r'hipt_env_on\("(HIPT_[A-Z0-9_]+)"\)', src)) | set(re.findall(r'getenv\("(HIPT_[A-Z0-9_]+)"\)', src)) | \
        set(re.findall(r'environ(?:\.get)?\(?\[?"(HIPT_[A-Z0-9_]+)"', src))
    missing_in_code = sorted(v for v in listed if v not in src)
    assert not missing_in_code, f"INTEGRATION.md lists switches nothing reads: {missing_in_code}"
    unlisted = sorted(v for v in read_by_lib if v not in listed)
    assert not unlisted, f"switches read by the code but missing from INTEGRATION.md section 6: {unlisted}"


def test_asm_read_audit_is_clean_and_part_of_the_build():
    """Every hot kernel relies on inline-asm loads whose destinations hipcc must not touch before a hand-counted wait (a hit is a
    wrong result or a GPU memory fault).  The Makefile audits the .s of every object as it is built and refuses to link on a hit;
    this test re-runs that verdict (make re-builds anything stale first) and checks the audit sees the in-flight reads at all."""
    import subprocess

    csrc = os.path.join(ROOT, "hipt_abmil_atec23_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j", "8", "audit"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for f in ("mlp16", "qkv_attention", "abmil32", "embed32", "seqgemm_pipe"):
        assert f"{f}.hip: violations: 0" in r.stdout, r.stdout
    # the checker itself: a register of an asm load still in flight across a loop back-edge, touched at the loop head
    syn = """_Z3fooPv:
.LBB0_1:
	v_mov_b32_e32 v9, v5
	;;#ASMSTART
	s_waitcnt vmcnt(0) ; XOP_FENCE
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[0:1], off
	;;#ASMEND
	s_cbranch_scc1 .LBB0_1
	;;#ASMSTART
	s_waitcnt vmcnt(0) ; XOP_FENCE
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        f.write(syn)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_asm_reads.py"), f.name], capture_output=True, text=True)
    os.unlink(f.name)
    assert r.returncode == 1 and "violations: 1" in r.stdout, r.stdout


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(N.NativeLibraryError, match="no non-HIP execution path"):
        N.lib()


def test_no_product_import_of_the_oracle():
    """The product package must never import oracle/ (test infrastructure only)."""
    pkg = os.path.join(ROOT, "hipt_abmil_atec23_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/ ", ""), os.path.join(dirpath, f)


def test_vit_state_dict_contract():
    from hipt_abmil_atec23_amd.vision_transformer import vit_small
    from hipt_abmil_atec23_amd.vision_transformer4k import vit4k_xs
    m = vit_small(patch_size=16, num_classes=0)
    spec = synth.vit_param_specs("vit256")
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: s for k, (s, _, _) in spec.items()}
    m4 = vit4k_xs(num_classes=0)
    spec4 = synth.vit_param_specs("vit4k", embed_dim=192, depth=6)
    assert {k: tuple(v.shape) for k, v in m4.state_dict().items()} == {k: s for k, (s, _, _) in spec4.items()}
    assert sum(p.numel() for p in m.parameters()) == 21_665_664
    assert sum(p.numel() for p in m4.parameters()) == 2_781_504
    # DINO 'teacher' dict with module./backbone. prefixes loads with strict=False (hipt_model_utils.py:61-70)
    sd = {"teacher": {"module.backbone." + k: v for k, v in synth.make_state_dict(spec).items()}}
    from hipt_abmil_atec23_amd import hipt_model_utils as U
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "vit256.pth")
        torch.save(sd, p)
        loaded = U.get_vit256(p)
    assert torch.equal(loaded.state_dict()["blocks.3.attn.qkv.weight"], synth.make_state_dict(spec)["blocks.3.attn.qkv.weight"])
    assert not any(q.requires_grad for q in loaded.parameters()) and not loaded.training
    with pytest.raises(AssertionError, match="pretrained weights not available"):
        U.get_vit256("/nonexistent/ckpt.pth")


def test_clam_state_dict_contract_and_demo_checkpoint_keys():
    from hipt_abmil_atec23_amd import CLAM_MB, CLAM_SB
    m = CLAM_SB(size_arg="hipt_big")
    assert list(m.state_dict().keys()) == list(synth.clam_param_specs((192, 128, 64)).keys())
    md = CLAM_SB(size_arg="small", dropout=0.25)
    assert "attention_net.3.attention_a.0.weight" in md.state_dict()
    # key layout of the reference's shipped demo checkpoint (heatmaps/demo/ckpts/s_0_checkpoint.pt), after the
    # cleaning of utils/eval_utils.py:51-57 ('.module' removed, 'instance_loss_fn*' dropped), loads with strict=True
    keys = {}
    for line in open(os.path.join(GOLDEN, "demo_ckpt_keys.txt")):
        k, shape = line.split(" ", 1)
        if "instance_loss_fn" in k:
            continue
        keys[k.replace(".module", "")] = tuple(int(x) for x in re.findall(r"\d+", shape))
    assert keys and set(keys) == set(md.state_dict().keys())
    md.load_state_dict({k: torch.zeros(s) for k, s in keys.items()}, strict=True)
    assert CLAM_SB(size_arg="hipt_384").attention_net[0].in_features == 384  # added size
    assert CLAM_SB(size_arg=[384, 128, 64]).classifiers.in_features == 128
    mb = CLAM_MB(size_arg="hipt_big", n_classes=3)
    assert len(mb.classifiers) == 3
    # xavier-normal weights / zero bias (utils/utils.py:217-225)
    assert float(m.classifiers.bias.abs().sum()) == 0.0 and float(m.classifiers.weight.std()) > 0.01
    for attr in ("k_sample", "n_classes", "subtyping", "instance_classifiers", "classifiers", "attention_net", "relocate"):
        assert hasattr(m, attr)


def test_cpu_inputs_vits_raise_clam_runs_torch_ops_on_the_cpu():
    from hipt_abmil_atec23_amd import CLAM_SB, HIPT_4K
    from hipt_abmil_atec23_amd.vision_transformer import vit_small
    with pytest.raises(RuntimeError, match="HIP device"):
        vit_small()(torch.zeros(1, 3, 256, 256))
    h = HIPT_4K(None, None, "cpu", "cpu")
    with pytest.raises(RuntimeError, match="HIP device"):
        h(torch.zeros(1, 3, 256, 256))
    # CLAM modules on CPU tensors run the PyTorch-op sequence on the CPU, as the reference does where relocate() finds no GPU
    # (models/model_clam.py:102-106); checked against the numpy oracle.  No native call is made.
    from hipt_abmil_atec23_amd import CLAM_MB, Attn_Net_Gated
    from oracle import hipt_oracle as O
    sc = synth.clam_param_specs((192, 128, 64))
    c = CLAM_SB(size_arg="hipt_big").eval()
    c.load_state_dict(synth.make_state_dict(sc, 7))
    hb = synth.hash_uniform_torch((40, 192), 11)
    calls = N.calls
    with torch.no_grad():
        logits, y_prob, y_hat, a_raw, _ = c(hb)
        a_only = c(hb, attention_only=True)
        g = Attn_Net_Gated(L=384, D=128).eval()
        a_g, x_g = g(torch.zeros(3, 384))
        mb = CLAM_MB(size_arg="hipt_big", n_classes=3).eval()
        out_mb = mb(hb)
    ref = O.clam_sb_forward(hb.numpy(), synth.make_params_np(sc, 7))
    assert np.abs(a_raw.numpy() - ref["A_raw"]).max() < 1e-5 and np.abs(logits.numpy() - ref["logits"]).max() < 1e-5
    assert int(y_hat) == int(np.asarray(ref["Y_hat"]).reshape(-1)[0]) and torch.equal(a_only, a_raw) and a_g.shape == (3, 1) and out_mb[0].shape == (1, 3)
    assert N.calls == calls
    # differentiable forward (main.py trains this module) is the documented PyTorch-ops training path
    logits, y_prob, y_hat, a_raw, res = c(torch.randn(20, 192), label=torch.tensor([1]), instance_eval=True)
    (logits.sum() + res["instance_loss"]).backward()
    assert c.classifiers.weight.grad is not None and a_raw.shape == (1, 20) and y_hat.dtype == torch.int64


def test_prepare_img_tensor_center_crop():
    from hipt_abmil_atec23_amd import HIPT_4K
    h = HIPT_4K(None, None, "cpu", "cpu")
    x = torch.arange(1 * 1 * 600 * 1000, dtype=torch.float32).reshape(1, 1, 600, 1000)
    img, w, hh = h.prepare_img_tensor(x)
    assert (w, hh) == (2, 3) and img.shape == (1, 1, 512, 768)
    assert torch.equal(img, x[:, :, 44:556, 116:884])  # int(round(88/2)) = 44, int(round(232/2)) = 116
    img2, w2, h2 = h.prepare_img_tensor(torch.zeros(1, 3, 512, 256))
    assert img2.shape == (1, 3, 512, 256) and (w2, h2) == (2, 1)


def test_dropin_module_paths():
    import sys
    import hipt_abmil_atec23_amd as amd
    from hipt_abmil_atec23_amd.dropin import uninstall
    amd.install()
    try:
        import HIPT_4K.hipt_4k as a
        import models.model_clam as b
        assert a.HIPT_4K is amd.HIPT_4K and b.CLAM_SB is amd.CLAM_SB
        assert "models" not in sys.modules or not hasattr(sys.modules["models"], "__file__") or True
    finally:
        uninstall()
    assert "HIPT_4K.hipt_4k" not in sys.modules and "models.model_clam" not in sys.modules


def test_eval_transforms_matches_totensor_normalize():
    from hipt_abmil_atec23_amd.hipt_model_utils import eval_transforms
    a = (np.arange(2 * 3 * 3, dtype=np.int64).reshape(2, 3, 3) * 14) % 256
    t = eval_transforms()(a.astype(np.uint8))
    ref = (torch.from_numpy(a.astype(np.float32) / 255.0).permute(2, 0, 1) - 0.5) / 0.5
    assert torch.allclose(t, ref)


def test_feature_store_roundtrip_and_subsampling(tmp_path):
    """pt_files/{slide}.pt writer (extract_features_fp.py:169-171,255) and the bag loader with max_patches_per_slide
    (datasets/dataset_generic.py:512-520: np.random.choice WITH replacement)."""
    from hipt_abmil_atec23_amd.feature_store import FeatureWriter, extract_slide, load_bag
    w = FeatureWriter(str(tmp_path), "slide_a", write_h5=False)
    f1, f2 = torch.randn(3, 192), torch.randn(2, 192)
    w.append(f1, torch.tensor([[0, 0], [0, 4096], [4096, 0]]))
    w.append(f2.numpy(), np.array([[4096, 4096], [8192, 0]]))
    assert len(w) == 5
    pt = w.close()
    assert pt.endswith(os.path.join("pt_files", "slide_a.pt"))
    stored = torch.load(pt)
    assert torch.is_tensor(stored) and stored.dtype == torch.float32 and torch.equal(stored, torch.cat([f1, f2]))
    assert torch.equal(load_bag(str(tmp_path), "slide_a"), stored)                   # no cap
    assert torch.equal(load_bag(str(tmp_path), "slide_a", 5), stored)                # cap not exceeded: untouched
    sub = load_bag(str(tmp_path), "slide_a", 3, rng=np.random.RandomState(0))
    idx = np.random.RandomState(0).choice(5, 3)
    assert sub.shape == (3, 192) and torch.equal(sub, stored[torch.as_tensor(idx)])
    many = load_bag(str(tmp_path), "slide_a", 4, rng=np.random.RandomState(7))       # with replacement: duplicates allowed
    assert many.shape == (4, 192)
    with pytest.raises(AssertionError, match="slide_missing"):
        load_bag(str(tmp_path), "slide_missing")
    with pytest.raises(ValueError):
        FeatureWriter(str(tmp_path), "bad", write_h5=False).append(torch.zeros(3, 192), torch.zeros(2, 2))
    # the driver loop with a stand-in model (any callable regions -> [R, d])
    model = lambda r: r.float().mean(dim=(1, 2, 3)).unsqueeze(1).repeat(1, 4)
    batches = [(torch.ones(2, 3, 8, 8) * k, torch.tensor([[k, 0], [k, 1]])) for k in (1, 2)]
    out = torch.load(extract_slide(model, batches, str(tmp_path), "slide_b"))
    assert out.shape == (4, 4) and torch.equal(out[:, 0], torch.tensor([1., 1., 2., 2.]))


def test_extract_slide_coalesces_batch_one_loader_batches(tmp_path):
    """The reference's loader yields one region per batch (extract_features_fp.py:128, 159-171); extract_slide gathers
    consecutive batches into calls of `coalesce` regions -- same files, same order, fewer calls; a batch of another shape or
    type is never merged."""
    from hipt_abmil_atec23_amd.feature_store import extract_slide
    calls = []

    def model(r):
        calls.append(tuple(r.shape))
        return r.float().mean(dim=(1, 2, 3)).unsqueeze(1).repeat(1, 4)

    def loader(n, shape=(3, 8, 8), dtype=torch.float32):
        return [(torch.full((1,) + shape, float(k), dtype=dtype), torch.tensor([[k, 7 * k]])) for k in range(n)]

    one = torch.load(extract_slide(model, loader(11), str(tmp_path), "one_by_one", coalesce=1))
    assert calls == [(1, 3, 8, 8)] * 11
    calls.clear()
    got = torch.load(extract_slide(model, loader(11), str(tmp_path), "gathered"))      # default: 8 regions per call
    assert calls == [(8, 3, 8, 8), (3, 3, 8, 8)] and torch.equal(got, one)
    calls.clear()
    mixed = loader(3) + loader(2, shape=(3, 4, 4)) + loader(2, dtype=torch.uint8) + loader(1, dtype=torch.uint8)
    out = torch.load(extract_slide(model, mixed, str(tmp_path), "mixed", coalesce=4))
    assert calls == [(3, 3, 8, 8), (2, 3, 4, 4), (3, 3, 8, 8)] and out.shape == (8, 4)
    assert torch.equal(out[:, 0], torch.tensor([0., 1., 2., 0., 1., 0., 1., 0.]))
    calls.clear()
    big = [(torch.ones(n, 3, 8, 8), torch.zeros(n, 2, dtype=torch.int64)) for n in (3, 6, 1)]
    torch.load(extract_slide(model, big, str(tmp_path), "big", coalesce=8))             # loader batches are never split
    assert calls == [(9, 3, 8, 8), (1, 3, 8, 8)]


def test_coords_survive_bit_for_bit(tmp_path):
    """coords [n, 2] are integer pass-through data (extract_features_fp.py:169-171, utils/file_utils.py:16-35).
    They are persisted on a host without h5py too (sidecar coords_files/{slide}.npy), in append order, in the dtype the loader
    handed over, bit for bit -- values above 2**53 included, which a detour through floats would destroy."""
    from hipt_abmil_atec23_amd.feature_store import FeatureWriter, coords_path, extract_slide, load_coords
    big = np.array([[2**62 + 1, -(2**61) - 3], [2**53 + 1, 7], [0, -1]], dtype=np.int64)
    w = FeatureWriter(str(tmp_path), "s64", write_h5=False)
    w.append(torch.zeros(3, 4), torch.from_numpy(big))          # a tensor batch ...
    w.append(np.ones((2, 4), np.float32), big[:2][::-1])         # ... and a (non-contiguous) numpy one
    w.close()
    got = load_coords(str(tmp_path), "s64")
    assert got.dtype == np.int64 and got.shape == (5, 2) and np.array_equal(got, np.concatenate([big, big[:2][::-1]]))
    assert got.tobytes() == np.concatenate([big, big[:2][::-1]]).tobytes()
    assert os.path.isfile(coords_path(str(tmp_path), "s64")) and sorted(os.listdir(tmp_path / "pt_files")) == ["s64.pt"]
    # the dtype of the loader survives (older patch files hold int32 coordinates); mixing dtypes or handing floats is an error
    w = FeatureWriter(str(tmp_path), "s32", write_h5=False)
    w.append(torch.zeros(2, 4), np.array([[1, 2], [3, 4]], dtype=np.int32))
    with pytest.raises(TypeError, match="int64"):
        w.append(torch.zeros(1, 4), np.array([[5, 6]], dtype=np.int64))
    w.close()
    assert load_coords(str(tmp_path), "s32").dtype == np.int32
    with pytest.raises(TypeError, match="integers"):
        FeatureWriter(str(tmp_path), "sf", write_h5=False).append(torch.zeros(1, 4), torch.zeros(1, 2))
    with pytest.raises(FileNotFoundError):
        load_coords(str(tmp_path), "never_written")
    # through the driver loop: one region per loader batch, gathered 8 per call with a ragged tail -- coordinates stay aligned
    # with the feature rows (the stand-in model writes a region's id into its feature, the loader the same id into its coords)
    model = lambda r: r.float().mean(dim=(1, 2, 3)).unsqueeze(1).repeat(1, 4)
    loader = [(torch.full((1, 3, 8, 8), float(k)), torch.tensor([[1000003 * k, -k]], dtype=torch.int64)) for k in range(19)]
    for name, co in (("co8", 8), ("co1", 1), ("co5", 5)):
        feats = torch.load(extract_slide(model, loader, str(tmp_path), name, coalesce=co))
        c = load_coords(str(tmp_path), name)
        assert c.dtype == np.int64 and c.shape == (19, 2)
        assert np.array_equal(c[:, 0], 1000003 * np.arange(19)) and np.array_equal(c[:, 1], -np.arange(19))
        assert torch.equal(feats[:, 0], torch.arange(19.))       # row i of the features <-> row i of the coordinates


def test_prepare_img_tensor_uint8_interleaved():
    from hipt_abmil_atec23_amd import HIPT_4K
    h = HIPT_4K(None, None, "cpu", "cpu")
    x = torch.randint(0, 256, (2, 600, 1000, 3), dtype=torch.uint8)
    img, w, hh = h.prepare_img_tensor(x)
    assert (w, hh) == (2, 3) and img.shape == (2, 512, 768, 3) and torch.equal(img, x[:, 44:556, 116:884, :])
    planar = x.permute(0, 3, 1, 2).contiguous()
    img2, _, _ = h.prepare_img_tensor(planar)
    assert img2.shape == (2, 3, 512, 768) and torch.equal(img2, planar[:, :, 44:556, 116:884])


def test_library_load_brings_torch_in_first():
    """Loading the library in a fresh interpreter must import torch (its bundled HIP runtime) before dlopen: kernels
    registered with the system runtime instead cannot be launched on torch's streams."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from hipt_abmil_atec23_amd import _native as N; "
            "assert 'torch' not in sys.modules; N.lib(); assert 'torch' in sys.modules; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def _replicate_like_data_parallel(module):
    """What torch.nn.parallel.replicate does to a module tree, on the CPU (the real one needs GPUs to broadcast to): replicas
    built by _replicate_for_data_parallel() have NO parameters(); the copies are plain tensor attributes, also listed in
    _former_parameters (torch/nn/parallel/replicate.py)."""
    from collections import OrderedDict
    mods = list(module.modules())
    idx = {m: i for i, m in enumerate(mods)}
    reps = []
    for m in mods:
        r = m._replicate_for_data_parallel()
        r._former_parameters = OrderedDict()
        reps.append(r)
    for i, m in enumerate(mods):
        for key, child in m._modules.items():
            setattr(reps[i], key, None if child is None else reps[idx[child]])
        for key, p in m._parameters.items():
            if p is not None:
                c = p.detach().clone()
                setattr(reps[i], key, c)
                reps[i]._former_parameters[key] = c
    return reps[0]


def test_hipt4k_survives_data_parallel_replication_and_deepcopy():
    """extract_features_fp.py:217-218 wraps the model in nn.DataParallel whenever it sees more than one GPU: a replica has
    no parameters(), so nothing on the forward path may rely on next(self.parameters())."""
    import copy
    import pickle

    from hipt_abmil_atec23_amd import HIPT_4K
    m = HIPT_4K(None, None, "cpu", "cpu").eval()
    r = _replicate_like_data_parallel(m)
    assert list(r.parameters()) == [] and list(r.model256.parameters()) == []
    assert r.model256.weight_device == torch.device("cpu") and r._same_device()
    assert len(r.model256._tensors()) == len(list(m.model256.parameters())) and len(r.model256._version_key()) > 100
    with pytest.raises(RuntimeError, match="HIP device"):  # the loud no-CPU-path error, not StopIteration
        r(torch.zeros(1, 3, 256, 256))
    # the device-side weight images are caches: deep copy / pickle of a module that has run must work and drop them
    class Unpicklable:
        def __reduce__(self):
            raise TypeError("ctypes objects containing pointers cannot be pickled")
    m.model256._packed[torch.device("cpu")] = ("key", Unpicklable())
    m2 = copy.deepcopy(m)
    assert m2.model256._packed == {} and m.model256._packed != {}
    assert torch.equal(m2.model256.pos_embed, m.model256.pos_embed)
    m3 = pickle.loads(pickle.dumps(m.model256))
    assert m3._packed == {} and torch.equal(m3.cls_token, m.model256.cls_token)


def test_vit_refuses_what_the_inference_kernels_would_get_wrong():
    from hipt_abmil_atec23_amd.vision_transformer import VisionTransformer
    m = VisionTransformer(embed_dim=64, depth=1, num_heads=2, drop_rate=0.1)
    m.train()
    with pytest.raises(RuntimeError, match="dropout"):
        m._check_inference_only()
    m.eval()
    with pytest.warns(UserWarning, match="no gradient flows"):
        m._check_inference_only()
    m._check_inference_only()  # said once
    qs = VisionTransformer(embed_dim=64, depth=1, num_heads=2, qk_scale=0.3)
    assert qs.blocks[0].attn.scale == 0.3  # travels to the kernels as hipt_vit_weights.attn_scale


def test_stream_argument_carries_its_device():
    import ctypes as C
    s = N.StreamArg(0)
    assert isinstance(s, C.c_void_p) and s.device is None
    t_cpu = torch.zeros(1)
    with pytest.raises(RuntimeError, match="expected all tensors on"):
        N.same_device("x", torch.device("cuda", 0), t_cpu)
    N.same_device("x", torch.device("cpu"), t_cpu, None)


def test_uint8_normalisation_constant_reproduces_every_byte_value():
    """csrc/embed32.hip normalises uint8 pixels in registers as bf16(fma(b, 2/255, -1)) with 2/255 = 0x3c008081 instead of the
    reference's ((b / 255) - 0.5) / 0.5 (ToTensor + Normalize, hipt_model_utils.py:113-118) rounded to bf16: the same bits for every
    one of the 256 inputs (a fused multiply-add is exact in float64 here: 8 x 24 significant bits, then one rounding to float32)."""
    import numpy as np
    import torch

    b = torch.arange(256, dtype=torch.float32)
    ref = ((b / 255) - 0.5) / 0.5
    s = np.array([0x3C008081], dtype=np.uint32).view(np.float32)[0]
    fused = (np.arange(256, dtype=np.float64) * np.float64(s) - 1.0).astype(np.float32)
    assert torch.equal(torch.from_numpy(fused).to(torch.bfloat16), ref.to(torch.bfloat16))
    # (and it is NOT the same float32: the kernel may only use it where the next step rounds to bf16)
    assert int((torch.from_numpy(fused) != ref).sum()) > 0


def test_feature_writer_h5_branch_with_a_stand_in_h5py(tmp_path, monkeypatch):
    """The `.h5` twin of the feature files (utils/file_utils.py:16-35: datasets `features` / `coords`, chunks (1, .), first axis resizable) is
    written only where h5py is importable -- no image of this project has it, so the branch had never executed (missing #4).
    Here a stand-in module with h5py's call surface (File as a context manager, create_dataset(name, data=, maxshape=, chunks=), item access)
    records what the writer asks for: the branch runs, with the reference's dataset names, chunking and resizable axis, and `load_coords` falls
    back to the .h5 when the sidecar is gone.  (A stand-in, not h5py: it pins the CALLS, not the file format.)"""
    import pickle
    import sys
    import types

    class _DS:
        def __init__(self, a):
            self.a = a

        def __getitem__(self, k):
            return self.a[k]

    class _File:
        def __init__(self, path, mode="r"):
            self.path, self.mode, self.d = path, mode, {}
            if mode == "r":
                self.d = pickle.load(open(path, "rb"))

        def __enter__(self):
            return self

        def __exit__(self, *a):
            if self.mode == "w":
                pickle.dump(self.d, open(self.path, "wb"))

        def create_dataset(self, name, data=None, maxshape=None, chunks=None, **kw):
            assert not kw, kw
            self.d[name] = {"data": np.array(data), "maxshape": maxshape, "chunks": chunks}

        def __getitem__(self, name):
            return _DS(self.d[name]["data"])

    fake = types.ModuleType("h5py")
    fake.File = _File
    monkeypatch.setitem(sys.modules, "h5py", fake)
    from hipt_abmil_atec23_amd.feature_store import FeatureWriter, coords_path, load_coords
    w = FeatureWriter(str(tmp_path), "slide_h5")  # write_h5=None: auto-detects the (stand-in) module
    assert w.write_h5
    f = torch.randn(5, 192)
    c = np.array([[0, 0], [0, 4096], [4096, 0], [4096, 4096], [2 ** 40, 7]], dtype=np.int64)
    w.append(f[:2], c[:2])
    w.append(f[2:], c[2:])
    w.close()
    h5 = os.path.join(str(tmp_path), "h5_files", "slide_h5.h5")
    d = pickle.load(open(h5, "rb"))
    assert set(d) == {"features", "coords"}
    assert np.array_equal(d["features"]["data"], f.numpy()) and d["features"]["data"].dtype == np.float32
    assert np.array_equal(d["coords"]["data"], c) and d["coords"]["data"].dtype == np.int64
    assert d["features"]["chunks"] == (1, 192) and d["features"]["maxshape"] == (None, 192)   # file_utils.py:24-28
    assert d["coords"]["chunks"] == (1, 2) and d["coords"]["maxshape"] == (None, 2)
    os.remove(coords_path(str(tmp_path), "slide_h5"))
    assert np.array_equal(load_coords(str(tmp_path), "slide_h5"), c)  # the reader's .h5 fallback


def test_wait_count_audits_catch_a_miscounted_wait(tmp_path):
    """tools/audit_ring_waits.py (embed32, seqgemm_pipe, mlp16) and tools/audit_qkv_wait.py are build gates (csrc/Makefile): the current listings pass, and a listing with a
    miscounted ring wait -- the round-6 defect: `vmcnt(12)` where hipcc had emitted eight loads -- or with a ninth output store fails."""
    import re
    import subprocess
    import sys
    build = os.path.join(ROOT, "hipt_abmil_atec23_amd", "csrc", "build")
    e32, qkv = os.path.join(build, "embed32.s"), os.path.join(build, "qkv_attention.s")
    if not (os.path.isfile(e32) and os.path.isfile(qkv)):
        pytest.skip("no device listings (the library was not built in this tree)")
    run = lambda tool, path: subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), path], capture_output=True, text=True)
    assert run("audit_ring_waits.py", e32).returncode == 0
    assert run("audit_qkv_wait.py", qkv).returncode == 0
    s = open(e32).read()
    # the interleaved-uint8 instantiation: every `vmcnt(8)` ring wait becomes the `vmcnt(12)` of the defective source
    i = s.index("_ZN12_GLOBAL__N_114embed32_kernelILi2ELb1EEEv11EmbedParams: ;")
    j = s.index("s_endpgm", i)
    bad = tmp_path / "embed32_bad.s"
    bad.write_text(s[:i] + s[i:j].replace("s_waitcnt vmcnt(8)", "s_waitcnt vmcnt(12)") + s[j:])
    r = run("audit_ring_waits.py", str(bad))
    assert r.returncode == 1 and "VIOLATION" in r.stdout and "vmcnt(12) before a ring barrier, but only 8 vector-memory instructions" in r.stdout
    for f in ("seqgemm_pipe.s", "mlp16.s"):  # the other two ring kernels' waits pass too
        assert run("audit_ring_waits.py", os.path.join(build, f)).returncode == 0, f
    q = open(qkv).read()
    m = re.search(r"\n(\s*buffer_store_dwordx2 [^\n]*)\n", q[q.index("_ZN12_GLOBAL__N_115qkv_attn_kernelILi0ELb0EEEvNS_13QkvAttnParamsE: ;"):])
    badq = tmp_path / "qkv_bad.s"
    k = q.index(m.group(1))
    badq.write_text(q[:k] + m.group(1) + "\n" + q[k:])  # a ninth output store
    r = run("audit_qkv_wait.py", str(badq))
    assert r.returncode == 1 and "expected exactly eight" in r.stdout
