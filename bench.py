#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: one JSON line on stdout from rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

``python bench.py --gpus N`` with N > 1 and no torchrun environment starts the N ranks ITSELF (a child
``python -m torch.distributed.run ...`` started before this process has touched the GPU; its output is relayed and its
exit code returned): a run never silently measures fewer GPUs than it was asked for (``n_gpus != --gpus`` is an error).

One STEP (per GPU) = one pass of the hot path over one batch of synthetic input, inputs resident in HBM before the timed
region:
  * HIPT_4K.forward on R resident [3,4096,4096] fp32 regions -> R x 256 patches through ViT-256 -> ViT-4K over each
    16x16 [CLS] grid -> [R,192]  (BASELINE.json configs[2], bf16 operands / fp32 accumulate);
  * CLAM_SB.forward (gated-attention pooling + bag classifier) on one 100 000 x 384 bf16 bag (configs[3]); bags rotate
    through a set larger than the 256 MiB Infinity Cache so every pass streams from HBM.
value = regions/s over ALL ranks (K steps per rank, max-over-ranks wall time between barriers); ``abmil_fwd_ms`` is the
HIP-event time of the CLAM_SB call inside the same timed steps (weak scaling: per-GPU work is fixed).

After the timed region:
  * ``config5`` (every N): BASELINE configs[4] -- 64 synthetic slides sharded slide i -> rank i mod N, per slide a stated
    SAMPLE of its ~8 192 regions through HIPT_4K (resident pixels), the slide's bag through CLAM_SB, ONE all-gather of all
    logits and ragged attention logits (``hipt_abmil_atec23_amd/pipeline.py``); slides/s extrapolated from the sample;
  * rank 0 re-runs a few steps with the library's per-kernel HIP-event hooks to price the kernels against their
    rooflines, measures the reference's own call patterns (one region per call; fp32 mode; one 256x256 patch in fp32 =
    configs[1]) and (N = 1 only) times the CPU oracle on the host cores (baseline only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "4K-regions/sec HIPT extract + ABMIL fwd ms @100k×384; 1/2/4/8 MI355X"
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
BAG_N, BAG_S0 = 100_000, 384
REGION = 4096
# 3 146 029 797 888 FLOP per region when every block runs on all 257 tokens.  The build runs the LAST
# ViT-256 block for the [CLS] query only (nothing else of it is consumed, vision_transformer.py:253); §8d requires the
# pruned figure then: per patch, block 12 keeps QKV (K and V of all tokens) and does QK^T / PV / proj / MLP for 1 of 257 rows.
# (round 2: Q of block 12 is computed for the [CLS] row only as well -- 256 of 257 rows of a [257, 384] x [384, 384] product less)
_BLOCK_TAIL = 50_725_632 + 50_725_632 + 75_792_384 + 606_339_072          # QK^T + PV + proj + MLP of one patch, one block
_Q_LAST = 2 * 257 * 384 * 384                                              # the Q third of block 12's QKV Linear, one patch
FLOP_PER_REGION = 3_146_029_797_888 - 256 * (_BLOCK_TAIL - _BLOCK_TAIL // 257) - 256 * (_Q_LAST - _Q_LAST // 257)
FLOP_PER_REGION_FULL = 3_146_029_797_888
LIB_CHUNK = 2048  # patches per ViT-256 pass when --chunk is 0 (capi.hip default_chunk)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=0, help="patches per ViT-256 pass (0 = library default, 2048)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the regions of a step are spread over (tail filling)")
    ap.add_argument("--u8", action="store_true", help="regions resident as uint8 RGB [R,W,H,3], normalised on the device (SURVEY 8f-1)")
    ap.add_argument("--regions", type=int, default=24, help="4096x4096 regions per HIPT_4K call (1 = the reference's batch_size)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the batch-1 / fp32 / single-patch figures")
    ap.add_argument("--slides", type=int, default=64, help="config 5: synthetic slides over all ranks (0 = skip the leg)")
    ap.add_argument("--slide-regions", type=int, default=8192, help="config 5: nominal regions per slide")
    ap.add_argument("--slide-sample", type=int, default=8, help="config 5: regions per slide actually extracted (stated sub-sample)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearsal of the multi-rank leg without a GPU: gloo backend, CPU tensors, stand-in models -- the same spawn / "
                         "init / barrier / timed loop / config 5 / gather / rank-0 JSON code path (tests/test_pipeline.py runs it with --gpus 2)")
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """--gpus N > 1 without a torchrun environment: start the N ranks as a CHILD job (this process has not initialised
    the GPU and never will: torch.cuda.device_count() does not) and relay it."""
    import socket

    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not args.dry_run:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_models():
    """--dry-run stand-ins with the call surface of HIPT_4K / CLAM_SB (CPU, deterministic): the driver code around them is the
    code under rehearsal, not they."""
    import torch

    def model(regions):
        r = regions.float().flatten(1)
        return r[:, :r.shape[1] // 192 * 192].reshape(r.shape[0], 192, -1).mean(dim=-1)

    def clam(bag):
        b = bag.float()
        a_raw = b.mean(dim=1).reshape(1, -1)
        m = torch.softmax(a_raw, dim=1) @ b
        logits = m[:, :2] * 3.0
        return logits, torch.softmax(logits, dim=1), logits.argmax(dim=1, keepdim=True), a_raw, {}

    model.streams = model.chunk = 0
    return model, clam, clam


def build_models(dev, dtype):
    from hipt_abmil_atec23_amd import CLAM_SB, HIPT_4K, synth
    m = HIPT_4K(None, None, dev, dev)
    m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
    m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
    m = m.eval().to(dev).set_compute_dtype(dtype)
    c = CLAM_SB(size_arg="hipt_384")
    c.load_state_dict(synth.make_state_dict(synth.clam_param_specs((384, 128, 64)), 384))
    c = c.eval().to(dev).set_compute_dtype(dtype)
    c192 = CLAM_SB(size_arg="hipt_big")  # the slide aggregator over HIPT_4K's 192-d region features (docs/README.md:69)
    c192.load_state_dict(synth.make_state_dict(synth.clam_param_specs((192, 128, 64)), 192))
    c192 = c192.eval().to(dev).set_compute_dtype(dtype)
    return m, c, c192


# What the chip SUSTAINS on this pool's boxes, measured by this repository's own probes -- the attainable bar beside the nominal one:
#   tools/mfma_shape_probe.hip (bare bf16 MFMA loops on random operands under the power cap): 2 220 TFLOP/s on 16x16x32 (2.35 GHz),
#   1 926 on 32x32x16 (1.9 GHz); tools/cu_bw_probe.hip: 6.2 TB/s with all 256 CUs streaming from HBM.
SUSTAINED_TFLOPS = {"16x16x32": 2220.0, "32x32x16": 1926.0}
SUSTAINED_HBM_GBS = 6200.0
MFMA_SHAPE_OF = {"mlp_fused": "16x16x32", "qkv_attention_fused": "32x32x16", "qkv_gemm": "16x16x32", "proj_gemm": "16x16x32", "attention": "16x16x32"}


SUSTAINED_SOURCE = ["constants typed from tools/mfma_shape_probe.hip runs of round 4 (DESIGN_HISTORY.md); profiles/r06_power.json holds round 6's measurement with W and MHz"]


def measure_sustained(dev, sampler=None, seconds=1.5):
    """The bf16 MFMA rate the chip SUSTAINS on THIS box in THIS run: bare 16x16x32 and 32x32x16 loops on random operands (tools/power_probe.hip, one
    wave per SIMD, 256 workgroups), `seconds` each after a 0.5 s ramp, HIP-event timed; with the hwmon sampler the W and MHz they ran at.
    Returns {shape: {...}} or None when the probe library is not built (build(): __graft_entry__.py)."""
    import ctypes as C

    import torch
    so = os.path.join(ROOT, "tools", "probe_bin", "libpower_probe.so")
    if not os.path.exists(so):
        return None
    pp = C.CDLL(so)
    pp.pp_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    pp.pp_flop_per_iter.restype, pp.pp_flop_per_iter.argtypes = C.c_double, [C.c_int]
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    outb = torch.zeros(2 * 4 * ncu, dtype=torch.int64, device=dev)
    iters = 200000  # 25.6 M matrix-pipe cycles per launch: ~12 ms
    res = {}
    for shape, name in ((16, "16x16x32"), (32, "32x32x16")):
        launch = lambda: pp.pp_launch(shape, ncu, iters, outb.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:
            launch()
            torch.cuda.synchronize()
        lo = sampler.mark() if sampler else 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(4):
                launch()
                n += 1
            torch.cuda.current_stream(dev).synchronize()
        b.record()
        torch.cuda.synchronize()
        tf = ncu * 4 * iters * pp.pp_flop_per_iter(shape) * n / (a.elapsed_time(b) * 1e-3) / 1e12
        h = outb.cpu().numpy().reshape(-1, 2).astype("float64")
        res[name] = {"tflops": tf, "in_kernel_ghz": float((h[:, 0] / h[:, 1]).mean() * 0.1)}
        if sampler:
            sm = sampler.summary(lo, sampler.mark(), skip_s=0.3)
            res[name].update({k: sm.get(k) for k in ("power_w_mean", "power_w_max", "sclk_mhz_mean")})
    return res


PROJ_FOLDED = [False]  # set by main(): the library runs the output projection inside the fused MLP kernel (image format 3, no HIPT_NO_PROJ_FOLD)


def kernel_flops(cat, rows, D=384, H=1536, heads=6, ntok=257):
    """Algorithmic FLOPs of one launch of a kernel category over `rows` token rows.  With the proj fold the fused MLP's
    launches also do the attention block's output projection (75 792 384 FLOP per patch and block): counted where it runs."""
    dh = D // heads
    att = 4 * (rows // ntok) * heads * ntok * ntok * dh
    return {"qkv_gemm": 2 * rows * 3 * D * D, "proj_gemm": 2 * rows * D * D,
            "fc1_gemm": 2 * rows * D * H, "fc2_gemm": 2 * rows * D * H, "mlp_fused": 4 * rows * D * H + (2 * rows * D * D if PROJ_FOLDED[0] else 0),
            "attention": att, "qkv_attention_fused": 2 * rows * 3 * D * D + att}.get(cat)


# rocprofv3 kernel names of the categories, most specific first (keys of profiles/<tag>_traffic.json)
TRAFFIC_KEYS = {"mlp_fused": ["void mlp16_kernel<true, true, 0, true>", "void mlp16_kernel<true, true, 0>", "void mlp32_kernel<true, true, 0>", "void mlp32_kernel<true, true, 0, false>"],
                "qkv_attention_fused": ["qkv_attn_kernel"],
                "qkv_gemm": ["void seqgemm_pipe_kernel<false, 0, true, true, false, true>", "void seqgemm_pipe_kernel<true, 0, true, false, false, false>",
                             "void seqgemm_pipe_kernel<true, 0>"],
                "proj_gemm": ["void seqgemm_pipe_kernel<false, 0, true, true, true, false>", "void seqgemm_pipe_kernel<false, 0, true, false, false, false>",
                              "void seqgemm_pipe_kernel<false, 0>"],
                "attention": ["attn64_kernel"],
                "abmil_fused": ["void abmil32_kernel<24>", "abmil_pipe_kernel", "void abmil_stream_kernel<6>"]}


def pmc_traffic(cat, with_source=False):
    """Per-launch HBM bytes from the newest committed PMC summary (separate rocprofv3 --pmc passes, corrected as
    MI355X_MICROARCH.md prescribes; tools/summarize_profile.py), or None.  NOT measured by this run (counters need rocprofv3):
    with_source=True also names the file the figure was read from and the commit that last touched it."""
    pdir = os.path.join(ROOT, "profiles")
    for f in sorted([p for p in os.listdir(pdir) if p.endswith("_traffic.json")], reverse=True) if os.path.isdir(pdir) else []:
        t = json.load(open(os.path.join(pdir, f)))
        meta = t.get("_meta") or {}
        for k in TRAFFIC_KEYS.get(cat, []):
            for name, v in t.items():
                if name != "_meta" and (name.startswith(k) or k in name):
                    if not with_source:
                        return v["hbm_bytes"]
                    commit = None
                    try:
                        import subprocess
                        commit = subprocess.run(["git", "-C", ROOT, "log", "-n", "1", "--format=%h", "--", os.path.join("profiles", f)],
                                                capture_output=True, text=True, timeout=10).stdout.strip() or None
                    except Exception:
                        pass
                    return v["hbm_bytes"], {"file": "profiles/" + f, "kernel": name, "commit": commit or meta.get("measured_at_commit"),
                                            "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this command (tools/run_profile.sh), per launch"}
    return (None, None) if with_source else None


def host_cores():
    """CPU share of this process: min(affinity, cgroup quota); the 1-GPU box grants 16 of its cores."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return min(n, int(os.environ.get("HIPT_BENCH_CPU_THREADS", "16")))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(check_bag_seed=None):
    """SURVEY.md §8(d) / BASELINE.md §4: the PyTorch-CPU restatement of the path (oracle/torch_cpu.py, pinned against the
    reference's outputs by tests/test_oracle_vs_golden.py) on the box's host cores, fp32, torch.set_num_threads(cores),
    warm-up 2 + min-of-5: ViT-256 on 16 patches (extrapolated to the 256 of a region) + one full ViT-4K; CLAM_SB on the
    2 000 x 384 and the 100 000 x 384 bag.  The numpy oracle's figure is kept as ``numpy_port``."""
    import numpy as np
    import torch

    from hipt_abmil_atec23_amd import synth
    from oracle import hipt_oracle as O
    from oracle import torch_cpu as TO
    cores = host_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cores)
    except Exception:
        pass

    def best(fn, warm=2, reps=5):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts)

    p256n = synth.make_params_np(synth.vit_param_specs("vit256"), 256)
    p4kn = synth.make_params_np(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096)
    pcn = synth.make_params_np(synth.clam_param_specs((384, 128, 64)), 384)
    p256, p4k, pc = TO.to_torch(p256n), TO.to_torch(p4kn), TO.to_torch(pcn)
    pos256 = torch.from_numpy(O.interpolate_pos_encoding(p256n["pos_embed"], 256, 256, 256, 16))
    pos4k = torch.from_numpy(O.interpolate_pos_encoding(p4kn["pos_embed"], 256, 16, 16, 1))
    n_p = 16
    with torch.no_grad():
        x = torch.from_numpy(synth.hash_uniform_np((n_p, 3, 256, 256), 3))
        t_p = best(lambda: TO.vit256_forward(x, p256, pos256))
        t_1 = best(lambda: TO.vit256_forward(x[:1], p256, pos256))
        g = torch.from_numpy(synth.hash_uniform_np((1, 384, 16, 16), 4))
        t_4k = best(lambda: TO.vit4k_forward(g, p4k, pos4k))
        bag = torch.from_numpy(synth.hash_uniform_np((BAG_N, BAG_S0), 4))
        t_2k = best(lambda: TO.clam_sb_forward(bag[:2000], pc))
        t_100k = best(lambda: TO.clam_sb_forward(bag, pc))
        # the checker's answer on the bag the GPU's last timed step pooled (rank 0, bag 40 + i): what `selfcheck` compares with
        chk = {}
        if check_bag_seed is not None:
            cb = torch.from_numpy(synth.hash_uniform_np((BAG_N, BAG_S0), check_bag_seed)).to(torch.bfloat16).float()  # (the resident bag is bf16)
            lg_, _, yh_, ar_, _ = TO.clam_sb_forward(cb, pc)
            chk = {"logits": lg_.reshape(-1).tolist(), "Y_hat": int(yh_.reshape(-1)[0]), "A_raw_head": ar_.reshape(-1)[:4096].tolist()}
    region_s = t_p / n_p * 256 + t_4k
    # the numpy oracle, one un-repeated bounded sample (round 1's figure, kept for continuity)
    xs = synth.hash_uniform_np((8, 3, 256, 256), 3)
    O.vit256_forward(xs[:2], p256n)
    t0 = time.perf_counter()
    O.vit256_forward(xs, p256n)
    t_np = (time.perf_counter() - t0) / 8 * 256
    torch.set_num_threads(prev)
    return {"value": 1.0 / region_s, "unit": "regions/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"PyTorch-CPU fp32 restatement (oracle/torch_cpu.py), {cores} threads, warm-up 2 + min of 5: ViT-256 on {n_p} of the "
                      f"256 patches of one 4096x4096 region ({t_p:.2f} s) extrapolated + one full ViT-4K ({t_4k * 1e3:.1f} ms); "
                      f"CLAM_SB on the full 2000x384 and 100000x384 fp32 bags",
            "patches_per_s": n_p / t_p, "seconds_per_region": region_s, "vit256_one_patch_ms": t_1 * 1e3,
            "abmil_fwd_ms": t_100k * 1e3, "abmil_2000_fwd_ms": t_2k * 1e3, "_check": chk,
            "numpy_port": {"regions_per_s": 1.0 / (t_np + t_4k), "sample": "numpy oracle, ViT-256 on 8 patches, one pass, extrapolated"}}


def emit_json_line(out) -> None:
    """The ONE line of this run, on the process's real stdout (see main: file descriptor 1 points at stderr while the run lasts)."""
    line = (json.dumps(out) + "\n").encode()
    os.write(_REAL_STDOUT[0] if _REAL_STDOUT else 1, line)


_REAL_STDOUT: list = []


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner to stdout when a communicator is
    # created): for the duration of the run file descriptor 1 is pointed at stderr, and the line goes to the saved descriptor.
    sys.stdout.flush()
    _REAL_STDOUT.append(os.dup(1))
    os.dup2(2, 1)

    import numpy as np
    import torch

    from hipt_abmil_atec23_amd import _native as N
    from hipt_abmil_atec23_amd import distributed as D
    from hipt_abmil_atec23_amd import pipeline as PL
    from hipt_abmil_atec23_amd import synth

    dry = args.dry_run
    rank, world, local = D.init_from_env(backend="gloo" if dry else None, single_rank_group=True)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the job has WORLD_SIZE={world} rank(s): refusing to report a number for a "
                         f"different GPU count than asked for")
    # a launcher (torch.distributed.run, also with ONE rank) means a live process group: every barrier / reduction / gather below is
    # then a real collective (RCCL on the GPU); plain `python bench.py` has none and they are local
    grp = [D.grouped()]
    group_info = D.group_info()
    if dry:
        # the rehearsal: every line of the multi-rank leg below runs, on CPU tensors over gloo, with stand-in models and small shapes
        dev = torch.device("cpu")
        model, clam, clam192 = dry_models()
        REGION_PX, n_bag = 32, 2000
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        N.lib()  # fail loudly before doing anything if the native library is missing
        model, clam, clam192 = build_models(dev, args.dtype)
        REGION_PX, n_bag = REGION, BAG_N
    model.streams = args.streams
    model.chunk = args.chunk
    R = args.regions
    region = synth.hash_uniform_torch((R, 3, REGION_PX, REGION_PX), 3 + rank, device=dev)
    if args.u8:  # the same pixels as decoded 8-bit RGB tiles (interleaved), 4x fewer bytes
        region = ((region * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    n_bags = 5  # 5 x 76.8 MB > 256 MiB Infinity Cache
    bag_dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    bags = [synth.hash_uniform_torch((n_bag, BAG_S0), 40 + 10 * rank + i, device=dev).to(bag_dt) for i in range(n_bags)]

    class _HostEvent:  # (--dry-run: the same bracketing, host clock)
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    new_event = _HostEvent if dry else (lambda: torch.cuda.Event(enable_timing=True))
    ev = [(new_event(), new_event()) for _ in range(args.steps)]

    def step(i, timed):
        f = model(region)
        if timed:
            ev[i][0].record()
        with torch.no_grad():
            logits, _, _, a_raw, _ = clam(bags[i % n_bags])
        if timed:
            ev[i][1].record()
        return f, logits, a_raw

    def barrier():
        if grp[0]:
            torch.distributed.barrier()
        if not dry:
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    # board power / shader clock over the timed steps: a thread of THIS process reading the card's hwmon node (tools/telemetry.py; sysfs only --
    # no GPU call, no child process), rank 0 only
    sampler = None
    if not dry and rank == 0 and os.environ.get("HIPT_BENCH_NO_POWER", "0") in ("", "0"):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from telemetry import PowerSampler, pci_bus_id_of
            sampler = PowerSampler(pci_bus_id_of(local), period_s=0.01).start()
        except Exception:
            sampler = None
    barrier()
    p_lo = sampler.mark() if sampler else 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(i, True)
    if grp[0]:  # the one collective of a job: per-slide logits + attention logits of every rank
        D.gather_slide_outputs([rank], [last[1]], [last[2]], world, device=dev)
    barrier()
    dt = time.perf_counter() - t0
    power_timed = sampler.summary(p_lo, sampler.mark()) if sampler else None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if grp[0]:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    abmil_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else None

    # ---- config 5 (BASELINE configs[4]): slides sharded over the ranks, one all-gather ----
    cfg5 = None
    if args.slides > 0:
        slides = PL.synthetic_slides(args.slides, args.slide_regions)
        ns = min(args.slide_sample, R)
        calls = [0]

        def resident(spec, idx):  # resident pixels: nothing is generated inside the timed region
            s = (calls[0] * ns) % max(1, R - ns + 1)
            calls[0] += 1
            return region[s:s + len(idx)]

        run5 = lambda: PL.process_slides(model, clam192, slides, rank, world, device=dev, regions_per_call=ns, sample_regions=ns,
                                         region_source=resident, expand_bag=True)
        run5()  # warm-up (first CLAM_SB [192,128,64] call, gather buffers)
        barrier()
        t5 = time.perf_counter()
        r5 = run5()
        barrier()
        t5_local = time.perf_counter() - t5  # (incl. the wait at the closing barrier: the job's time as this rank saw it)
        d5 = torch.tensor([t5_local], dtype=torch.float64, device=dev)
        nreg = torch.tensor([r5.local_regions], dtype=torch.float64, device=dev)
        # per-rank load: regions, slides and the seconds a rank spent in its own slides (before the gather) -- min / max over the
        # ranks show the imbalance of the slide-level sharding directly
        mine = torch.tensor([float(r5.local_regions), float(len(r5.local_slides)), float(r5.local_seconds)],
                            dtype=torch.float64, device=dev)
        lo, hi = mine.clone(), mine.clone()
        if grp[0]:
            torch.distributed.all_reduce(d5, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(nreg, op=torch.distributed.ReduceOp.SUM)
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        d5, nreg = float(d5.item()), float(nreg.item())
        per_rank = {"regions_min": int(lo[0].item()), "regions_max": int(hi[0].item()), "slides_min": int(lo[1].item()),
                    "slides_max": int(hi[1].item()), "seconds_min": float(lo[2].item()), "seconds_max": float(hi[2].item())}
        total_regions = sum(s.n_regions for s in slides)
        cfg5 = {"slides": args.slides, "regions_per_slide_nominal": args.slide_regions, "regions_sampled_per_slide": ns,
                "seconds": d5, "regions_per_s": nreg / d5, "sampled_slides_per_s": args.slides / d5,
                "slides_per_s_extrapolated": args.slides / (d5 * total_regions / nreg),
                "per_rank": per_rank,
                "note": "a PLUMBING check of BASELINE configs[4], not a throughput figure: slide i -> rank i mod G; per slide only the stated "
                        "sample of its regions goes through HIPT_4K (bf16), the features are tiled to the slide's n regions, CLAM_SB "
                        "[192,128,64] and ONE all-gather of logits + ragged A_raw carry their true sizes.  slides_per_s_extrapolated = "
                        "measured time x (all regions / sampled regions): it charges the per-slide CLAM_SB + gather at the sampled rate and "
                        "is dominated by the 8 sampled regions per slide -- the headline `value` is the measured throughput",
                "gathered_logits_shape": list(r5.logits.shape), "gathered_a_raw_total": int(sum(a.numel() for a in r5.a_raw))}

    if grp[0] and world == 1:
        # a one-rank group has done its work (timed gather, config 5): end it here, so that the single-GPU legs below run as in a
        # plain launch (a HIP-graph capture with RCCL's watchdog thread alive is not safe)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        grp[0] = False

    # the CLAM kernel's own pace: K forwards captured in ONE HIP graph (the arrival ticket of the in-kernel merge returns to zero by
    # itself, so the launch replays) and replayed between ONE HIP-event pair on the stream they run on -- no host time between the
    # launches, no event record per launch.  Rotating bags (5 x 77 MB > Infinity Cache).
    abmil_graph = None
    # (single-process runs only: with a process group alive, RCCL's watchdog thread may touch the runtime while a capture is open)
    if not dry and args.steps and world == 1 and not args.no_extras:
        try:
            KG, REP = 20, 5
            gs = torch.cuda.Stream()
            with torch.no_grad(), torch.cuda.stream(gs):
                for i in range(3):
                    clam(bags[i % n_bags])
            gs.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(graph, stream=gs):
                keep = [clam(bags[i % n_bags]) for i in range(KG)]
            graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(REP):
                with torch.cuda.stream(gs):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(gs)
                    graph.replay()
                    b.record(gs)
                gs.synchronize()
                ts.append(a.elapsed_time(b) / KG * 1e3)
            with torch.no_grad():
                same = bool(torch.equal(keep[0][0], clam(bags[0])[0]))
            abmil_graph = {"us_per_launch_mean": float(np.mean(ts)), "us_per_launch_min": float(np.min(ts)), "launches": KG, "replays": REP,
                           "same_logits_as_a_plain_call": same}
            del graph, keep
        except Exception as e:  # a graph that cannot be captured on this stack is reported, not fatal
            abmil_graph = {"error": repr(e)[:200]}

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()  # rank 0's single-GPU legs below: keep the group alive until it is done
            torch.distributed.destroy_process_group()
        return

    if dry:
        out = {"metric": METRIC, "value": world * args.steps * R / dt, "unit": "regions/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "dry_run": True,
               "config": {"workload": "DRY RUN (no GPU, not a measurement): stand-in models on CPU tensors over gloo through the code path of the "
                                      "multi-rank leg", "regions_per_step": R, "parallelism": f"slide-sharded x{world}, one all-gather"},
               "abmil_fwd_ms": abmil_ms}
        if cfg5:
            out["config5"] = cfg5
        if group_info:
            out["group_ranks"], out["collective_backend"] = group_info["ranks"], group_info["backend"]
        emit_json_line(out)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    chunk = args.chunk or LIB_CHUNK
    per_stream = -(-R // max(1, min(args.streams, R))) * 256
    out = {
        "metric": METRIC, "value": world * args.steps * R / dt, "unit": "regions/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "per step: HIPT_4K.forward on R resident [3,4096,4096] fp32 regions (R x 256 patches ViT-256 -> "
                               "ViT-4K over each 16x16 [CLS] grid, BASELINE configs[2]) + CLAM_SB gated-attention pooling over one "
                               "100000x384 bag (configs[3], the bag resident in the compute dtype); random-init weights of the "
                               "reference architectures",
                   "regions_per_step": R, "streams": args.streams, "input": "uint8 RGB interleaved" if args.u8 else "fp32 normalised",
                   "bag": [BAG_N, BAG_S0], "bag_dtype": "bf16-resident" if args.dtype == "bf16" else "fp32-resident",
                   "chunk_patches": min(chunk, per_stream), "parallelism": f"slide-sharded x{world}, one all-gather"},
        "abmil_fwd_ms": abmil_ms,
        "model_tflops": world * args.steps * R * FLOP_PER_REGION / dt / 1e12,
    }
    if cfg5:
        out["config5"] = cfg5
    if group_info:
        # the process group the timed gather and config 5 ran over (backend "nccl" = RCCL on ROCm), as the group itself reports it
        out["rccl_ranks" if group_info["backend"] == "nccl" else "group_ranks"] = group_info["ranks"]
        out["collective_backend"] = group_info["backend"]

    if power_timed is not None:
        # what the board drew and the shader clock it held over the timed steps (hwmon power1_input / freq1_input, 10 ms samples)
        out["power"] = dict(power_timed, note="socket power and sclk over the timed steps; power_cap_w = hwmon power1_cap (the board limit)")
    # the ceilings `frac_of_sustained` is quoted against, measured by THIS run on THIS box (bare MFMA loops, 2 s each) -- or the typed constants
    if not args.no_extras and world == 1:
        try:
            ms_ = measure_sustained(dev, sampler)
            if ms_:
                for shp, v in ms_.items():
                    SUSTAINED_TFLOPS[shp] = v["tflops"]
                SUSTAINED_SOURCE[0] = "measured by this run: bare bf16 MFMA loops on random operands, one wave per SIMD, 256 workgroups (tools/power_probe.hip)"
                out["sustained_mfma"] = ms_
        except Exception as e:
            out["sustained_mfma"] = {"error": repr(e)[:200]}
    if sampler is not None:
        sampler.stop()  # (the legs below are host-bound in places: no second Python thread beside them)

    # ---- per-kernel roofline leg: same workload, HIP events around every launch, ONE stream ----
    prof = {}
    if args.profile_steps > 0:
        model.streams = 1  # per-launch HIP events: one stream, so that a kernel's time is its own (no other kernel on the CUs)
        N.profile_enable(True)
        for i in range(args.profile_steps):
            step(i, False)
        torch.cuda.synchronize()
        prof = N.profile_read()
        N.profile_enable(False)
        model.streams = args.streams
    kernels = {}
    for cat, (ms, cnt) in prof.items():
        kernels[cat] = {"launches_per_step": cnt / args.profile_steps, "total_ms": ms, "avg_us": ms / cnt * 1e3,
                        "ms_per_step": ms / args.profile_steps}
    out["kernels"] = kernels
    # every launch of these categories is a full-size one (the [CLS]-row launches of the pruned last block are booked
    # under 'last_block_cls'): rows per launch = one chunk of patches x 257, or all of the step's patches when fewer
    try:
        pk256 = model.model256._packed_for(model.model256._pos_for(256, 256, 256))
        PROJ_FOLDED[0] = N.lib().hipt_vit_mlp_pack_format(pk256.ref) == 3 and os.environ.get("HIPT_NO_PROJ_FOLD", "0") in ("", "0")
    except Exception:
        PROJ_FOLDED[0] = False
    rows_launch = min(chunk, 256 * R) * 257
    mf = {c: v for c, v in kernels.items() if kernel_flops(c, rows_launch)}
    if PROJ_FOLDED[0]:
        mf.pop("proj_gemm", None)  # (what is left in that category are the second-level ViT's small launches: no full-size proj launch exists)
    flops_of = lambda c: kernel_flops(c, rows_launch)
    if mf and (256 * R) % min(chunk, 256 * R) == 0:
        frac = {}
        for c, v in mf.items():
            ach = flops_of(c) / (v["avg_us"] * 1e-6) / 1e12
            frac[c] = {"achieved": ach, "frac": ach / PEAK_TFLOPS[args.dtype], "avg_launch_us": v["avg_us"], "launches_per_step": v["launches_per_step"]}
        dom = max(mf, key=lambda c: mf[c]["ms_per_step"])  # dominant kernel = largest share of the step
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": frac[dom]["achieved"], "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                           "frac": frac[dom]["frac"], "mfma_shape": MFMA_SHAPE_OF.get(dom), "peak_sustained": SUSTAINED_TFLOPS.get(MFMA_SHAPE_OF.get(dom)),
                           "frac_of_sustained": (frac[dom]["achieved"] / SUSTAINED_TFLOPS[MFMA_SHAPE_OF[dom]]) if args.dtype == "bf16" and dom in MFMA_SHAPE_OF else None,
                           "peak_sustained_source": SUSTAINED_SOURCE[0],
                           "traffic": pmc_traffic(dom), "traffic_source": pmc_traffic(dom, True)[1], "flops_per_launch": flops_of(dom),
                           "rows_per_launch": rows_launch, "launches_per_step": mf[dom]["launches_per_step"], "avg_launch_us": mf[dom]["avg_us"],
                           "proj_folded_into_fused_mlp": PROJ_FOLDED[0]}
        # (only the categories whose every launch is a full-size one: the others also hold the second-level ViT's small launches, whose
        #  average time says nothing about a 2 048-patch launch)
        out["roofline_all"] = {c: round(f["frac"], 4) for c, f in frac.items() if c in ("mlp_fused", "qkv_attention_fused")}
        if PROJ_FOLDED[0] and "qkv_attention_fused" in frac and "mlp_fused" in frac:
            # north_star's "ViT-256 attention" unit with the output projection folded into the fused MLP kernel: the fused QKV + attention kernel,
            # the side GEMM over the [CLS] rows, and proj's share of the fused MLP launch -- its share of that launch's FLOPs, i.e. proj is
            # priced at the pace of the kernel it runs in (6 of its 54 ring phases)
            side = kernels.get("qkv_cls_rows", {}).get("ms_per_step", 0.0) / kernels["qkv_attention_fused"]["launches_per_step"] * 1e3
            fl_proj = 2 * rows_launch * 384 * 384
            share = fl_proj / kernel_flops("mlp_fused", rows_launch)
            proj_us = frac["mlp_fused"]["avg_launch_us"] * share
            us = frac["qkv_attention_fused"]["avg_launch_us"] + side + proj_us
            fl = kernel_flops("qkv_attention_fused", rows_launch) + fl_proj
            out["roofline_attention_unit"] = {"kernels": ["qkv_attention_fused", "qkv_cls_rows", "mlp_fused (proj share)"], "bound": "mfma", "us_per_launch_set": us,
                                              "cls_rows_us_per_block": side, "proj_share_of_fused_mlp": share, "proj_us_inside_fused_mlp": proj_us,
                                              "achieved": fl / (us * 1e-6) / 1e12, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                              "frac": fl / (us * 1e-6) / 1e12 / PEAK_TFLOPS[args.dtype], "target": 0.60,
                                              "mfma_shape": "32x32x16", "peak_sustained": SUSTAINED_TFLOPS["32x32x16"],
                                              "frac_of_sustained": fl / (us * 1e-6) / 1e12 / SUSTAINED_TFLOPS["32x32x16"],
                                              "traffic": pmc_traffic("qkv_attention_fused"),
                                              "traffic_source": pmc_traffic("qkv_attention_fused", True)[1]}
        elif all(c in frac for c in ("qkv_attention_fused", "proj_gemm")):
            # north_star's "ViT-256 attention" unit (LN1 + QKV + QK^T + softmax + PV + proj = 404 620 800 FLOP per patch and block) as the
            # LayerNorm-chained blocks run it: the fused QKV + attention kernel, the side GEMM over the [CLS] rows (two small launches,
            # booked as 'qkv_cls_rows': their time per block is added), the proj GEMM
            side = kernels.get("qkv_cls_rows", {}).get("ms_per_step", 0.0) / kernels["qkv_attention_fused"]["launches_per_step"] * 1e3
            us = frac["qkv_attention_fused"]["avg_launch_us"] + side + frac["proj_gemm"]["avg_launch_us"]
            fl = kernel_flops("qkv_attention_fused", rows_launch) + kernel_flops("proj_gemm", rows_launch)
            out["roofline_attention_unit"] = {"kernels": ["qkv_attention_fused", "qkv_cls_rows", "proj_gemm"], "bound": "mfma", "us_per_launch_set": us,
                                              "cls_rows_us_per_block": side,
                                              "achieved": fl / (us * 1e-6) / 1e12, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                              "frac": fl / (us * 1e-6) / 1e12 / PEAK_TFLOPS[args.dtype], "target": 0.60,
                                              "mfma_shape": "32x32x16", "peak_sustained": SUSTAINED_TFLOPS["32x32x16"],
                                              "frac_of_sustained": fl / (us * 1e-6) / 1e12 / SUSTAINED_TFLOPS["32x32x16"],
                                              "traffic": pmc_traffic("qkv_attention_fused"),
                                              "traffic_source": pmc_traffic("qkv_attention_fused", True)[1]}
        elif all(c in frac for c in ("qkv_gemm", "attention", "proj_gemm")):
            # north_star's "ViT-256 attention" unit: LN1 + QKV + QK^T + softmax + PV + proj = 404 620 800 FLOP per patch and block
            us = sum(frac[c]["avg_launch_us"] for c in ("qkv_gemm", "attention", "proj_gemm"))
            fl = sum(kernel_flops(c, rows_launch) for c in ("qkv_gemm", "attention", "proj_gemm"))
            out["roofline_attention_unit"] = {"kernels": ["qkv_gemm", "attention", "proj_gemm"], "bound": "mfma", "us_per_launch_set": us,
                                              "achieved": fl / (us * 1e-6) / 1e12, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                              "frac": fl / (us * 1e-6) / 1e12 / PEAK_TFLOPS[args.dtype], "target": 0.60}
    if "abmil_fused" in kernels:
        esz = 2 if args.dtype == "bf16" else 4
        alg = BAG_N * BAG_S0 * esz + 4 * BAG_N
        # `frac` / `achieved` / `avg_launch_us`: one HIP-event pair around every launch, as every other roofline entry of this line and
        # as in earlier rounds.  The kernel's own pace (20 launches in one replayed HIP graph between ONE event pair: no event record
        # between launches) is reported beside it under its own names -- single-process runs with extras only.
        us = kernels["abmil_fused"]["avg_us"]
        gbs = alg / (us * 1e-6) / 1e9
        ra = {"kernel": "abmil_fused", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
              "frac": gbs / PEAK_HBM_GBS, "peak_sustained": SUSTAINED_HBM_GBS, "frac_of_sustained": gbs / SUSTAINED_HBM_GBS,
              "traffic": pmc_traffic("abmil_fused"), "traffic_source": pmc_traffic("abmil_fused", True)[1],
              "algorithmic_bytes": alg, "avg_launch_us": us, "timing": "one HIP-event pair around every launch (library profile)",
              "graph": abmil_graph, "target": 0.50}
        if abmil_graph and "us_per_launch_mean" in abmil_graph:
            ug = abmil_graph["us_per_launch_mean"]
            ra.update({"avg_launch_us_graph": ug, "achieved_graph": alg / (ug * 1e-6) / 1e9, "frac_graph": alg / (ug * 1e-6) / 1e9 / PEAK_HBM_GBS,
                       "frac_graph_of_sustained": alg / (ug * 1e-6) / 1e9 / SUSTAINED_HBM_GBS,
                       "timing_graph": (f"{abmil_graph['launches']} launches in one replayed HIP graph between one HIP-event pair on their "
                                        f"stream, mean of {abmil_graph['replays']} replays")})
        out["roofline_abmil"] = ra

    # ---- the reference's own call patterns (SURVEY.md §8d configs 2-3 as the unmodified scripts issue them) ----
    if not args.no_extras:
        def timed(fn, n, warm=2):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        ex = {}
        one = region[:1]
        # extract_features_fp.py:159-171: batch_size 1.  HIPT_4K's own default for ONE region (patch_streams = 2: the patches of the region over two
        # streams) is what an unmodified reference loop gets; beside it the one-stream form
        model.streams, model.patch_streams = 2, 2
        ex["batch1_regions_per_s"] = 1.0 / timed(lambda: model(one), 10)           # (HIPT_4K's default since round 6: the region's patches over two streams)
        model.patch_streams = 1
        ex["batch1_one_patch_stream_regions_per_s"] = 1.0 / timed(lambda: model(one), 10)
        model.patch_streams = 2
        if region.shape[0] >= 8:
            # the same one-region loader batches through feature_store.extract_slide's loop, which gathers 8 of them per call
            # (same files bit for bit: tests/test_gpu_parity.py::test_extract_slide_gathered_calls_write_the_same_bits)
            import tempfile
            from hipt_abmil_atec23_amd.feature_store import extract_slide
            model.streams = args.streams
            loader = [(region[i:i + 1], torch.tensor([[4096 * i, 0]])) for i in range(8)] * 2
            with tempfile.TemporaryDirectory() as td:
                extract_slide(model, loader[:8], td, "warm")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                extract_slide(model, loader, td, "timed")
                torch.cuda.synchronize()
                ex["batch1_loader_gathered_regions_per_s"] = len(loader) / (time.perf_counter() - t0)  # incl. the .pt write
            model.streams = 1
        if not args.u8:
            model.set_compute_dtype("fp32")  # the reference's own precision (exact-fp32 MFMA kernels, 1e-4 parity mode)
            t32 = timed(lambda: model(one), 2, warm=1)
            ex["fp32_regions_per_s"] = 1.0 / t32
            ex["fp32_model_tflops"] = FLOP_PER_REGION_FULL / t32 / 1e12
            ex["fp32_frac_of_fp32_mfma_peak"] = ex["fp32_model_tflops"] / PEAK_TFLOPS["fp32"]
            patch = synth.hash_uniform_torch((1, 3, 256, 256), 2, device=dev)
            ex["config2_vit256_one_patch_fp32_ms"] = timed(lambda: model.model256(patch), 20) * 1e3  # BASELINE configs[1]
            model.set_compute_dtype(args.dtype)
            ex["config2_vit256_one_patch_bf16_ms"] = timed(lambda: model.model256(patch), 20) * 1e3
        if not args.u8 and args.dtype == "bf16":
            # the same regions as decoded 8-bit RGB tiles (interleaved [R, W, H, 3]: a quarter of the bytes over the host link;
            # normalised inside the patch-embedding kernel, bit-identical features): the production input format
            model.streams = args.streams
            reg8 = ((region * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            ex["uint8_input_regions_per_s"] = R / timed(lambda: model(reg8), 5)
            # ---- the loop WITH its H2D hop (extract_features_fp.py:162-166): one-region loader batches in PINNED HOST memory through
            # feature_store.extract_slide (8 per call, copy stream + double-buffered gather buffers + features read back one call late), beside the
            # same loop over resident batches.  uint8 RGB tiles: 50 MB per region over the link; fp32: 201 MB (the link alone caps that at ~300 / s)
            try:
                import tempfile
                from hipt_abmil_atec23_amd.feature_store import extract_slide
                h2d = {}
                for kind, src, reps in (("uint8", reg8[:8], 12), ("fp32", region[:8], 3)):  # (12 / 3 gathered calls: the first call's copy is the only one nothing hides)
                    res_loader = [(src[i:i + 1], torch.tensor([[4096 * i, 0]])) for i in range(8)] * reps
                    pinned = [src[i:i + 1].cpu().pin_memory() for i in range(8)]
                    host_loader = [(pinned[i], torch.tensor([[4096 * i, 0]])) for i in range(8)] * reps
                    with tempfile.TemporaryDirectory() as td:
                        for name, ld in (("resident", res_loader), ("host", host_loader)):
                            extract_slide(model, ld[:8], td, "warm")
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            extract_slide(model, ld, td, "timed")
                            torch.cuda.synchronize()
                            h2d[f"{kind}_{name}"] = len(ld) / (time.perf_counter() - t0)
                    del pinned, host_loader, res_loader
                ex["h2d_uint8_regions_per_s"] = h2d["uint8_host"]
                ex["h2d_fp32_regions_per_s"] = h2d["fp32_host"]
                ex["h2d_loop_resident_uint8_regions_per_s"] = h2d["uint8_resident"]
                ex["h2d_loop_resident_fp32_regions_per_s"] = h2d["fp32_resident"]
                ex["h2d_uint8_over_resident"] = h2d["uint8_host"] / h2d["uint8_resident"]
                ex["h2d_note"] = ("extract_slide over one-region loader batches, 8 gathered per call, incl. the .pt / coords write: `host` = batches in pinned host "
                                  "memory (copy stream, two gather buffers), `resident` = the same loop over batches already in HBM")
            except Exception as e:
                ex["h2d_error"] = repr(e)[:300]
            # ... and the uint8 loop in a FRESH process (a child: this process keeps the GPU; nothing is exec'd over it), where the copy stream does not
            # share a hardware queue with a dozen older streams -- what an extraction job looks like (tools/h2d_loop_bench.py)
            try:
                # (192 regions, 24 gathered per call: a slide is thousands of regions -- the first call's copy, which nothing hides, is 3 % here)
                cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "h2d_loop_bench.py"), "24", "24"], capture_output=True, text=True, timeout=240)
                line = [l for l in cp.stdout.splitlines() if l.startswith("H2D_LOOP ")]
                ex["h2d_fresh_process"] = json.loads(line[-1][len("H2D_LOOP "):]) if line else {"error": (cp.stderr or cp.stdout)[-300:]}
            except Exception as e:
                ex["h2d_fresh_process"] = {"error": repr(e)[:300]}
            del reg8
            model.streams = 1
        # What an event pair adds to ONE launch (why `abmil_fwd_ms` / roofline_abmil.avg_launch_us read ~4 us above the graph-replayed pace of
        # the same kernel): pairs around a one-element elementwise kernel, enqueued behind a long call so that the host is never the limiter.
        # The reading = event record + dependent-dispatch latency + a ~1 us kernel; a launch inside a graph pays none of it.
        try:
            tiny = torch.zeros(1, device=dev)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            model(one)
            for a_, b_ in evs:
                a_.record()
                tiny.add_(1.0)
                b_.record()
            torch.cuda.synchronize()
            ex["event_pair_around_trivial_kernel_us"] = float(np.median([a_.elapsed_time(b_) for a_, b_ in evs])) * 1e3
        except Exception as e:
            ex["event_pair_around_trivial_kernel_us"] = repr(e)[:100]
        bag2k = synth.hash_uniform_torch((2000, BAG_S0), 1, device=dev)
        clam.set_compute_dtype("fp32")
        with torch.no_grad():
            ex["clam_sb_2000x384_fp32_ms"] = timed(lambda: clam(bag2k), 20) * 1e3  # BASELINE configs[0] on the GPU
        clam.set_compute_dtype(args.dtype)
        model.streams = args.streams
        out["extras"] = ex

    # ---- selfcheck: the timed workload's own outputs against the committed golden vector / the CPU checker ----
    sc = {}
    if args.dtype == "bf16" and not args.u8 and R >= 1:
        try:  # region 0 of rank 0 = seed 3 = the input of tests/golden/hipt4k_4096.npz (the reference's own output on it)
            g = np.load(os.path.join(ROOT, "tests", "golden", "hipt4k_4096.npz"))["out"].astype(np.float64).ravel()
            f0 = last[0][0].float().cpu().numpy().astype(np.float64).ravel()
            sc["region0_rel_l2_vs_reference_golden"] = float(np.linalg.norm(f0 - g) / np.linalg.norm(g))
            sc["region0_ok"] = bool(sc["region0_rel_l2_vs_reference_golden"] <= 1.3e-2)  # the bf16 bar of tests/test_gpu_parity.py
        except Exception as e:  # a missing fixture must not eat the bench line
            sc["region0_error"] = repr(e)
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(check_bag_seed=40 + (args.steps - 1) % n_bags if args.steps else None)
        chk = cb.pop("_check", {})
        out["cpu_baseline"] = cb
        if chk and args.dtype == "bf16":  # the last timed step's bag through the PyTorch-CPU restatement (bf16-rounded bag, fp32 math)
            lg = last[1].float().cpu().numpy().ravel()
            ar = last[2].float().cpu().numpy().ravel()[:len(chk["A_raw_head"])]
            sc["bag_logits_max_abs"] = float(np.abs(lg - np.asarray(chk["logits"])).max())
            sc["bag_a_raw_max_abs"] = float(np.abs(ar - np.asarray(chk["A_raw_head"])).max())
            sc["bag_ok"] = bool(sc["bag_logits_max_abs"] <= 1e-3 and sc["bag_a_raw_max_abs"] <= 4e-2 and int(np.argmax(lg)) == chk["Y_hat"])
    else:
        out.pop("_check", None)
    oks = [v for k, v in sc.items() if k.endswith("_ok")]
    out["selfcheck"] = ("ok" if oks and all(oks) else ("FAILED" if oks else "skipped"))
    out["selfcheck_detail"] = sc
    emit_json_line(out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
