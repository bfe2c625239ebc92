#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X (contract: one JSON line on stdout from rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP (per GPU) = one pass of the hot path over one batch of synthetic input, inputs resident in
HBM before the timed region:
  * HIPT_4K.forward on one [1,3,4096,4096] fp32 region -> 256 patches through ViT-256 -> ViT-4K over
    the 16x16 [CLS] grid -> [1,192]  (BASELINE.json configs[2], bf16 operands / fp32 accumulate);
  * CLAM_SB.forward (gated-attention pooling + bag classifier) on one 100 000 x 384 bf16 bag
    (configs[3]); bags rotate through a set larger than the 256 MiB Infinity Cache so every pass
    streams from HBM.
value = regions/s over ALL ranks (K steps per rank, max-over-ranks wall time between barriers);
``abmil_fwd_ms`` is the HIP-event time of the CLAM_SB call inside the same timed steps.  With N > 1
the steps are sharded "slide"-wise with no data-path collective and end with the one all-gather of
per-slide logits / attention logits (weak scaling: per-GPU work is fixed).

After the timed region rank 0 re-runs a few steps with the library's per-kernel HIP-event hooks to
price the dominant kernel against its roofline, and (N = 1 only) times the numpy oracle on the host
cores as the CPU baseline (a bounded sample, extrapolated; baseline only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "4K-regions/sec HIPT extract + ABMIL fwd ms @100k×384; 1/2/4/8 MI355X"
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
BAG_N, BAG_S0 = 100_000, 384
REGION = 4096
# 3 146 029 797 888 FLOP per region when every block runs on all 257 tokens.  The build runs the LAST
# ViT-256 block for the [CLS] query only (nothing else of it is consumed, vision_transformer.py:253); §8d requires the
# pruned figure then: per patch, block 12 keeps QKV (K and V of all tokens) and does QK^T / PV / proj / MLP for 1 of 257 rows.
_BLOCK_TAIL = 50_725_632 + 50_725_632 + 75_792_384 + 606_339_072          # QK^T + PV + proj + MLP of one patch, one block
FLOP_PER_REGION = 3_146_029_797_888 - 256 * (_BLOCK_TAIL - _BLOCK_TAIL // 257)


def build_models(dev, dtype):
    from hipt_abmil_atec23_amd import CLAM_SB, HIPT_4K, synth
    m = HIPT_4K(None, None, dev, dev)
    m.model256.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
    m.model4k.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096))
    m = m.eval().to(dev).set_compute_dtype(dtype)
    c = CLAM_SB(size_arg="hipt_384")
    c.load_state_dict(synth.make_state_dict(synth.clam_param_specs((384, 128, 64)), 384))
    c = c.eval().to(dev).set_compute_dtype(dtype)
    return m, c


def kernel_flops(cat, rows, D=384, H=1536, heads=6, ntok=257):
    """Algorithmic FLOPs of one launch of a kernel category over `rows` token rows."""
    dh = D // heads
    return {"qkv_gemm": 2 * rows * 3 * D * D, "proj_gemm": 2 * rows * D * D,
            "fc1_gemm": 2 * rows * D * H, "fc2_gemm": 2 * rows * D * H, "mlp_fused": 4 * rows * D * H,
            "attention": 4 * (rows // ntok) * heads * ntok * ntok * dh}.get(cat)


def step_flops(cat, rows, passes, ntok=257):
    """Algorithmic FLOPs of a category over one step: `passes` chunks of `rows` token rows through 12 blocks, the last
    block pruned to the [CLS] row for everything after QKV (1 of ntok rows; attention: 1 of ntok queries)."""
    full = kernel_flops(cat, rows)
    if full is None:
        return None
    return passes * (12 * full if cat == "qkv_gemm" else 11 * full + full // ntok)


# rocprofv3 kernel names of the categories, most specific first (keys of profiles/<tag>_traffic.json; template
# arguments of the pipelined kernels: <LN, dbg, packed weights, A image, out image, head-major out> / <dbg, packed, images, x-in image>)
TRAFFIC_KEYS = {"mlp_fused": ["void mlp_pipe_kernel<0, true, true, true>", "void mlp_pipe_kernel<0, true, false, false>", "void mlp_pipe_kernel<0>"],
                "qkv_gemm": ["void seqgemm_pipe_kernel<false, 0, true, true, false, true>", "void seqgemm_pipe_kernel<true, 0, true, false, false, false>",
                             "void seqgemm_pipe_kernel<true, 0>"],
                "proj_gemm": ["void seqgemm_pipe_kernel<false, 0, true, true, true, false>", "void seqgemm_pipe_kernel<false, 0, true, false, false, false>",
                              "void seqgemm_pipe_kernel<false, 0>"],
                "attention": ["attn64_kernel"],
                "abmil_fused": ["abmil_pipe_kernel", "void abmil_stream_kernel<6>"]}


def pmc_traffic(cat):
    """Per-launch HBM bytes from the committed PMC summary (separate rocprofv3 --pmc passes, corrected
    as MI355X_MICROARCH.md prescribes; tools/summarize_profile.py), or None."""
    for f in sorted([p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_traffic.json")], reverse=True) \
            if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
        t = json.load(open(os.path.join(ROOT, "profiles", f)))
        for k in TRAFFIC_KEYS.get(cat, []):
            if k in t:
                return t[k]["hbm_bytes"]
    return None


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


def cpu_baseline(budget_s=12.0):
    """numpy-oracle ("port") timing on the host cores: ViT-256 on a bounded sample of patches of one
    region (+ the full ViT-4K), extrapolated to regions/s; CLAM_SB on the full 100k x 384 bag."""
    from hipt_abmil_atec23_amd import synth
    from oracle import hipt_oracle as O
    cores = host_cores()
    try:  # keep the BLAS pool inside the box's CPU share (oversubscription makes numpy crawl)
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cores)
    except Exception:
        pass
    p256 = synth.make_params_np(synth.vit_param_specs("vit256"), 256)
    p4k = synth.make_params_np(synth.vit_param_specs("vit4k", embed_dim=192, depth=6), 4096)
    pc = synth.make_params_np(synth.clam_param_specs((384, 128, 64)), 384)
    x = synth.hash_uniform_np((8, 3, 256, 256), 3)
    O.vit256_forward(x[:2], p256)  # warm-up (BLAS thread pool, page-in)
    t0 = time.perf_counter()
    O.vit256_forward(x, p256)
    t8 = time.perf_counter() - t0
    n = int(max(8, min(256, 8 * (budget_s * 0.8 / max(t8, 1e-3)) // 8 * 8)))
    xs = synth.hash_uniform_np((n, 3, 256, 256), 3)
    t0 = time.perf_counter()
    for i in range(0, n, 16):
        O.vit256_forward(xs[i:i + 16], p256)
    t_patches = time.perf_counter() - t0
    g = synth.hash_uniform_np((1, 384, 16, 16), 4)
    t0 = time.perf_counter()
    O.vit4k_forward(g, p4k)
    t4k = time.perf_counter() - t0
    region_s = t_patches / n * 256 + t4k
    bag = synth.hash_uniform_np((BAG_N, BAG_S0), 4)
    O.clam_sb_forward(bag[:1000], pc)
    t0 = time.perf_counter()
    O.clam_sb_forward(bag, pc)
    tb = time.perf_counter() - t0
    return {"value": 1.0 / region_s, "unit": "regions/s", "cores": cores, "kind": "port",
            "sample": f"numpy fp32 oracle: ViT-256 on {n} of the 256 patches of one 4096x4096 region "
                      f"({t_patches:.2f} s) + full ViT-4K ({t4k * 1e3:.0f} ms), extrapolated to one region; "
                      f"CLAM_SB on the full 100000x384 fp32 bag",
            "abmil_fwd_ms": tb * 1e3, "seconds_per_region": region_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--chunk", type=int, default=0, help="patches per ViT-256 pass (0 = library default)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the regions of a step are spread over (tail filling)")
    ap.add_argument("--u8", action="store_true", help="regions resident as uint8 RGB [R,W,H,3], normalised on the device (SURVEY 8f-1)")
    ap.add_argument("--regions", type=int, default=24, help="4096x4096 regions per HIPT_4K call (1 = the reference's batch_size)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from hipt_abmil_atec23_amd import _native as N
    from hipt_abmil_atec23_amd import distributed as D
    from hipt_abmil_atec23_amd import synth

    rank, world, local = D.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    N.lib()  # fail loudly before doing anything if the native library is missing

    model, clam = build_models(dev, args.dtype)
    model.streams = args.streams
    model.chunk = args.chunk
    R = args.regions
    region = synth.hash_uniform_torch((R, 3, REGION, REGION), 3 + rank, device=dev)
    if args.u8:  # the same pixels as decoded 8-bit RGB tiles (interleaved), 4x fewer bytes
        region = ((region * 0.5 + 0.5) * 255).round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    n_bags = 5  # 5 x 76.8 MB > 256 MiB Infinity Cache
    bag_dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    bags = [synth.hash_uniform_torch((BAG_N, BAG_S0), 40 + 10 * rank + i, device=dev).to(bag_dt) for i in range(n_bags)]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    feats, slide_logits, last_a = [], [], None

    def step(i, timed):
        nonlocal last_a
        f = model(region)
        if timed:
            ev[i][0].record()
        with torch.no_grad():
            logits, _, _, a_raw, _ = clam(bags[i % n_bags])
        if timed:
            ev[i][1].record()
            feats.append(f)
            slide_logits.append(logits)
        last_a = a_raw

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    if world > 1:  # the one collective of the job: per-slide logits + attention logits of every rank
        D.gather_slide_outputs([rank], [slide_logits[-1]], [last_a], world, device=dev)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    abmil_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank != 0:
        return
    out = {
        "metric": METRIC, "value": world * args.steps * R / dt, "unit": "regions/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "per step: HIPT_4K.forward on R resident [3,4096,4096] fp32 regions (R x 256 patches ViT-256 -> "
                               "ViT-4K over each 16x16 [CLS] grid, BASELINE configs[2]) + CLAM_SB gated-attention pooling over one "
                               "100000x384 bag (configs[3]); random-init weights of the reference architectures",
                   "regions_per_step": R, "streams": args.streams, "input": "uint8 RGB interleaved" if args.u8 else "fp32 normalised", "bag": [BAG_N, BAG_S0], "chunk_patches": args.chunk or 256 * R,
                   "parallelism": f"slide-sharded x{world}, one all-gather"},
        "abmil_fwd_ms": abmil_ms,
        "model_tflops": world * args.steps * R * FLOP_PER_REGION / dt / 1e12,
    }

    # ---- per-kernel roofline leg: same workload, HIP events around every launch ----
    prof = {}
    if args.profile_steps > 0:
        model.streams = 1  # per-launch HIP events: one stream, so that a kernel's time is its own (no other kernel on the CUs)
        N.profile_enable(True)
        for i in range(args.profile_steps):
            step(i, False)
        torch.cuda.synchronize()
        prof = N.profile_read()
        N.profile_enable(False)
    chunk_rows = (args.chunk or 256 * R) * 257
    passes = (256 * R) // (args.chunk or 256 * R)  # ViT-256 passes (chunks of patches) per step
    kernels = {}
    for cat, (ms, cnt) in prof.items():
        kernels[cat] = {"launches_per_step": cnt / args.profile_steps, "total_ms": ms, "avg_us": ms / cnt * 1e3,
                        "ms_per_step": ms / args.profile_steps}
    out["kernels"] = kernels
    mf = {c: v for c, v in kernels.items() if kernel_flops(c, chunk_rows)}
    if mf:
        # dominant kernel = largest share of the step.  Each category holds the 12 ViT-256 launches of a step (11 over
        # R x 65 792 rows, the last block's over the R x 256 [CLS] rows): achieved = their algorithmic FLOPs / their
        # HIP-event time; avg_launch_us is the average over the 12.
        dom = max(mf, key=lambda c: mf[c]["ms_per_step"])
        fl_step = step_flops(dom, chunk_rows, passes)  # (the ViT-4K launches are booked under 'vit4k_blocks')
        ach = fl_step / (mf[dom]["ms_per_step"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                           "frac": ach / PEAK_TFLOPS[args.dtype], "traffic": pmc_traffic(dom),
                           "flops_per_step": fl_step, "launches_per_step": mf[dom]["launches_per_step"],
                           "avg_launch_us": mf[dom]["avg_us"]}
        out["roofline_all"] = {c: round(step_flops(c, chunk_rows, passes) /
                                        (v["ms_per_step"] * 1e-3) / 1e12 / PEAK_TFLOPS[args.dtype], 4) for c, v in mf.items()}
    if "abmil_fused" in kernels:
        esz = 2 if args.dtype == "bf16" else 4
        alg = BAG_N * BAG_S0 * esz + 4 * BAG_N
        gbs = alg / (kernels["abmil_fused"]["avg_us"] * 1e-6) / 1e9
        out["roofline_abmil"] = {"kernel": "abmil_fused", "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": gbs / PEAK_HBM_GBS, "traffic": pmc_traffic("abmil_fused"), "algorithmic_bytes": alg,
                                 "avg_launch_us": kernels["abmil_fused"]["avg_us"]}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
