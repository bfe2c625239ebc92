"""Slide-level driver (BASELINE configs[4] / SURVEY.md §8d config 5, §8e): sharding, feature store, per-slide CLAM_SB,
the one all-gather.  CPU part: two gloo ranks with stand-in callables (the driver holds no device code).  GPU part
(-m gpu): the real HIP models at world size 1 against the numpy oracle, and at world size 2 over RCCL when two GPUs exist."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PX = 512  # region side of the test slides: 2 x 2 patches of 256 -> a 2 x 2 [CLS] grid for ViT-4K


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ---- stand-ins with the call surface of HIPT_4K / CLAM_SB (CPU, deterministic) ----
def fake_model(regions):
    r = regions.float()
    return torch.stack([r.mean(dim=(1, 2, 3)), r.amax(dim=(1, 2, 3)), r[:, 0].mean(dim=(1, 2)), r[:, :, 0, 0].sum(dim=1)], dim=1)


def fake_clam(bag):
    a_raw = (bag * torch.tensor([1.0, -2.0, 0.5, 0.25])).sum(dim=1).reshape(1, -1)
    m = torch.softmax(a_raw, dim=1) @ bag
    logits = m[:, :2] * 3.0
    return logits, torch.softmax(logits, dim=1), logits.argmax(dim=1, keepdim=True), a_raw, {}


def _slides():
    from hipt_abmil_atec23_amd import pipeline as PL
    return [PL.SlideSpec(f"s{i}", n_regions=3 + (i * 5) % 4, seed=i, region_px=32, grid_cols=4) for i in range(5)]


def _cpu_worker(rank, world, port, feat_dir, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from hipt_abmil_atec23_amd import distributed as D
    from hipt_abmil_atec23_amd import pipeline as PL
    r, w, _ = D.init_from_env(backend="gloo")
    run = PL.process_slides(fake_model, fake_clam, _slides(), r, w, device=torch.device("cpu"), regions_per_call=2, feat_dir=feat_dir)
    q.put((rank, run.local_slides, run.local_regions, run.logits.numpy(), [a.numpy() for a in run.a_raw]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(180)
def test_process_slides_two_gloo_ranks_equal_one_process(tmp_path):
    from hipt_abmil_atec23_amd import feature_store as FS
    from hipt_abmil_atec23_amd import pipeline as PL
    slides = _slides()
    ref = PL.process_slides(fake_model, fake_clam, slides, 0, 1, device=torch.device("cpu"), regions_per_call=2, keep_features=True)
    assert ref.local_slides == [0, 1, 2, 3, 4] and ref.local_regions == sum(s.n_regions for s in slides)
    assert [a.numel() for a in ref.a_raw] == [s.n_regions for s in slides]  # ragged
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]  # slide i -> rank i mod G
    assert res[0][2] + res[1][2] == ref.local_regions
    for rk in res:  # every rank holds every slide's outputs, equal to the one-process run bit for bit
        assert np.array_equal(rk[3], ref.logits.numpy())
        assert all(np.array_equal(a, b.numpy()) for a, b in zip(rk[4], ref.a_raw))
    for i, s in enumerate(slides):  # the feature store holds each slide's [n, d] features (extract_features_fp.py:248-255)
        f = FS.load_bag(str(tmp_path), s.slide_id)
        assert torch.equal(f, ref.local_features[i].float())
        c = FS.load_coords(str(tmp_path), s.slide_id)  # ... and its coords [n, 2], int64, row for row (extract_features_fp.py:169-171)
        assert c.dtype == np.int64 and np.array_equal(c, s.coords(range(s.n_regions)).numpy())


def test_process_slides_sampling_skip_existing_and_coords(tmp_path):
    from hipt_abmil_atec23_amd import pipeline as PL
    slides = _slides()
    assert PL.sample_indices(10, None) == list(range(10)) and PL.sample_indices(10, 4) == [0, 2, 5, 7] and PL.sample_indices(3, 8) == [0, 1, 2]
    c = slides[0].coords([0, 1, 5])
    assert c.dtype == torch.int64 and c.tolist() == [[0, 0], [32, 0], [32, 32]]
    calls = []

    def counting_model(x):
        calls.append(x.shape[0])
        return fake_model(x)

    a = PL.process_slides(counting_model, fake_clam, slides, device=torch.device("cpu"), sample_regions=2, expand_bag=True, feat_dir=str(tmp_path))
    assert a.local_regions == 2 * len(slides) and [x.numel() for x in a.a_raw] == [s.n_regions for s in slides]  # bags tiled back to n
    n_calls = len(calls)
    b = PL.process_slides(counting_model, fake_clam, slides, device=torch.device("cpu"), sample_regions=2, expand_bag=True,
                          feat_dir=str(tmp_path), skip_existing=True)
    assert len(calls) == n_calls and b.local_regions == 0  # auto-skip: nothing re-extracted (extract_features_fp.py:231-238)
    assert torch.equal(a.logits, b.logits)
    sl = PL.synthetic_slides(64, 8192)
    assert len(sl) == 64 and all(7168 <= s.n_regions <= 8192 for s in sl) and len({s.n_regions for s in sl}) > 1


@pytest.mark.timeout(300)
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_dry_run_two_ranks(launcher):
    """bench.py's multi-rank leg rehearsed without a GPU (--dry-run: gloo, CPU tensors, stand-in models): spawn_ranks (or the
    driver's own `python -m torch.distributed.run ...` form) -> init_from_env -> barriers + timed loop -> config 5 over 64
    slides -> the all-gather -> ONE JSON line from rank 0.  The first contact of that code path with more than one rank must
    not be the 8-GPU node."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    tail = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"]
    if launcher == "self":
        cmd = [sys.executable, bench] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 alone reports
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config5"]["slides"] == 64 and d["config5"]["gathered_logits_shape"] == [64, 2]
    assert d["config5"]["gathered_a_raw_total"] == sum(s.n_regions for s in __import__("hipt_abmil_atec23_amd.pipeline", fromlist=["x"]).synthetic_slides(64, 8192))
    # a job with another world size than asked for refuses to report
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), bench, "--gpus", "4", "--dry-run", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert bad.returncode != 0 and not any(l.startswith("{") for l in bad.stdout.splitlines())


def test_bench_dry_run_eight_ranks_and_fewer_slides_than_ranks():
    """The driver's 8-rank launch form rehearsed on CPU / gloo: (i) 8 ranks over the default 64 slides, (ii) 8 ranks over 5
    slides -- three ranks have NO slide: they must contribute an empty (all -1) block to the gather and still reach every barrier,
    and the gathered result must be the whole job.  config5's per-rank min / max show the imbalance directly."""
    import json
    import subprocess
    from hipt_abmil_atec23_amd import pipeline as PL
    bench = os.path.join(ROOT, "bench.py")
    for slides in (64, 5):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), bench, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run", "--slides", str(slides)]
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")  # 8 ranks on the container's 8 cores
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=560, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        c5 = d["config5"]
        assert d["dry_run"] is True and d["n_gpus"] == 8 and d["value"] > 0
        assert c5["slides"] == slides and c5["gathered_logits_shape"] == [slides, 2]
        assert c5["gathered_a_raw_total"] == sum(s.n_regions for s in PL.synthetic_slides(slides, 8192))
        pr = c5["per_rank"]
        if slides == 5:
            assert pr["slides_min"] == 0 and pr["slides_max"] == 1 and pr["regions_min"] == 0 and pr["regions_max"] > 0
        else:
            assert pr["slides_min"] == pr["slides_max"] == 8 and pr["regions_min"] > 0
        assert 0 <= pr["seconds_min"] <= pr["seconds_max"]


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the real models
# ---------------------------------------------------------------------------------------------------------------------
def _gpu_models(dev):
    from hipt_abmil_atec23_amd import CLAM_SB, HIPT_4K, synth
    m = HIPT_4K(None, None, dev, dev)
    m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
    m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
    m = m.eval().to(dev)
    c = CLAM_SB(size_arg="hipt_big")
    c.load_state_dict(synth.make_state_dict(synth.clam_param_specs((192, 128, 64)), 192))
    return m, c.eval().to(dev)


def _gpu_slides():
    from hipt_abmil_atec23_amd import pipeline as PL
    return [PL.SlideSpec(f"g{i}", n_regions=2 + i % 3, seed=10 + i, region_px=PX) for i in range(4)]


def _oracle_slides(slides):
    from hipt_abmil_atec23_amd import pipeline as PL
    from hipt_abmil_atec23_amd import synth
    from oracle import hipt_oracle as O
    p256 = synth.make_params_np(synth.vit_param_specs("vit256"), 256)
    p4k = synth.make_params_np(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096)
    pc = synth.make_params_np(synth.clam_param_specs((192, 128, 64)), 192)
    out = []
    for s in slides:
        x = PL.hashed_regions(s, range(s.n_regions), "cpu").numpy()
        f = np.concatenate([O.hipt4k_forward(x[i:i + 1], p256, p4k) for i in range(s.n_regions)], axis=0)
        r = O.clam_sb_forward(f, pc)
        out.append((f, r["logits"].reshape(-1), r["A_raw"].reshape(-1)))
    return out


def _check_against_oracle(logits, a_raw, feats, slides):
    ref = _oracle_slides(slides)
    for i, (f, lg, ar) in enumerate(ref):
        if i in feats:
            assert np.abs(feats[i] - f).max() < 1e-4, i     # region features (fp32 mode: 1e-4)
        assert np.abs(logits[i] - lg).max() < 1e-4, i       # slide logits
        assert a_raw[i].shape == ar.shape and np.abs(a_raw[i] - ar).max() < 1e-4, i   # attention logits, ragged


@pytest.mark.gpu
def test_config5_pipeline_world1_matches_oracle(tmp_path):
    from hipt_abmil_atec23_amd import _native as N
    from hipt_abmil_atec23_amd import pipeline as PL
    dev = torch.device("cuda:0")
    m, c = _gpu_models(dev)
    slides = _gpu_slides()
    before = N.calls
    run = PL.process_slides(m, c, slides, 0, 1, device=dev, regions_per_call=2, keep_features=True, feat_dir=str(tmp_path))
    assert N.calls > before  # the HIP library did the work
    assert run.local_slides == [0, 1, 2, 3] and run.logits.shape == (4, 2)
    _check_against_oracle(run.logits.cpu().numpy(), [a.cpu().numpy() for a in run.a_raw],
                          {i: f.cpu().numpy() for i, f in run.local_features.items()}, slides)
    # bf16 (the bench configuration) through the same driver: the stated bf16 bar on the slide logits
    m.set_compute_dtype("bf16")
    c.set_compute_dtype("bf16")
    run16 = PL.process_slides(m, c, slides, 0, 1, device=dev, regions_per_call=3)
    ref = run.logits.cpu().numpy()
    rel = float(np.linalg.norm(run16.logits.cpu().numpy() - ref) / np.linalg.norm(ref))
    print(f"config-5 pipeline bf16 vs fp32 slide logits: rel-L2 {rel:.2e}")
    assert rel < 2e-2


def _nccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from hipt_abmil_atec23_amd import distributed as D
    from hipt_abmil_atec23_amd import pipeline as PL
    r, w, local = D.init_from_env(backend="nccl")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    m, c = _gpu_models(dev)
    run = PL.process_slides(m, c, _gpu_slides(), r, w, device=dev, regions_per_call=2)
    q.put((rank, run.local_slides, run.logits.cpu().numpy(), [a.cpu().numpy() for a in run.a_raw]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _nccl_one_rank_worker(port, q):
    """ONE rank under the launcher's environment: init_from_env creates a real 1-rank RCCL communicator, and the driver's
    all-reduce + all-gather run over it on DEVICE tensors (no local-copy shortcut: distributed.grouped())."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from hipt_abmil_atec23_amd import distributed as D
    from hipt_abmil_atec23_amd import pipeline as PL
    r, w, local = D.init_from_env(single_rank_group=True)
    info = D.group_info()
    calls = {"all_reduce": 0, "all_gather": 0}
    real_ar, real_ag = torch.distributed.all_reduce, torch.distributed.all_gather_into_tensor

    def ar(t, *a, **k):
        calls["all_reduce"] += int(t.is_cuda)
        return real_ar(t, *a, **k)

    def ag(o, t, *a, **k):
        calls["all_gather"] += int(t.is_cuda and o.is_cuda)
        return real_ag(o, t, *a, **k)

    torch.distributed.all_reduce, torch.distributed.all_gather_into_tensor = ar, ag
    dev = torch.device("cuda", local)
    m, c = _gpu_models(dev)
    run = PL.process_slides(m, c, _gpu_slides(), r, w, device=dev, regions_per_call=2, keep_features=True)
    q.put((info, calls, run.local_slides, run.logits.cpu().numpy(), [a.cpu().numpy() for a in run.a_raw],
           {i: f.cpu().numpy() for i, f in run.local_features.items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_config5_pipeline_one_rank_rccl_matches_oracle():
    """First RCCL contact on the hardware at hand: the slide driver + gather_slide_outputs over a ONE-rank
    process group with backend "nccl" (= RCCL), device tensors through the collectives, outputs against the numpy oracle as in
    test_config5_pipeline_world1_matches_oracle.  In a child process: the group must not outlive the test."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_one_rank_worker, args=(_free_port(), q))
    p.start()
    info, calls, local_slides, logits, a_raw, feats = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert info == {"backend": "nccl", "ranks": 1}
    assert calls["all_reduce"] >= 1 and calls["all_gather"] == 1, calls  # the shape agreement + THE one all-gather, on the device
    assert local_slides == [0, 1, 2, 3] and logits.shape == (4, 2)
    _check_against_oracle(logits, a_raw, feats, _gpu_slides())


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_one_rank_through_the_launcher_reports_rccl_ranks():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (the driver's launch form, at the one GPU there is):
    the same JSON line as a plain launch plus "rccl_ranks" taken from the live process group; the timed gather and config 5 ran
    over RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--regions", "3", "--profile-steps", "0",
           "--no-cpu-baseline", "--no-extras", "--slides", "4", "--slide-sample", "2"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=850, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert r.stdout.strip() == lines[0], "stdout must hold the JSON line and nothing else (RCCL prints a version banner to fd 1): " + r.stdout[:300]
    assert d["rccl_ranks"] == 1 and d["collective_backend"] == "nccl" and d["n_gpus"] == 1 and d["value"] > 0
    assert d["config5"]["gathered_logits_shape"] == [4, 2]
    assert d["selfcheck"] in ("ok", "skipped"), d.get("selfcheck_detail")


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_config5_pipeline_world2_rccl():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL all-gather over xGMI); the 1-GPU box runs the world-1 form")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]
    for rk in res:
        _check_against_oracle(rk[2], rk[3], {}, _gpu_slides())
