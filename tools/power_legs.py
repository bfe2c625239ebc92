"""Board power and shader clock per leg (round 6, VERDICT r5 #3: "measure the power cap instead of inferring it").

One process: the main thread drives the GPU, a sampler thread reads the board's hwmon node (tools/telemetry.py; sysfs only).  Legs, each held
for --seconds (the first --skip seconds of a leg are dropped: the SMU's socket-power reading is a moving average):
    idle | bare 16x16x32 bf16 MFMA loop | bare 32x32x16 loop (tools/power_probe.hip, one wave per SIMD, random operands) |
    the fused proj + MLP kernel alone (hipt_vit_mlp_unit, 2 048 patches) | the fused QKV + attention kernel alone (hipt_vit_attention_unit) |
    the full bench step (HIPT_4K.forward on 24 resident regions over 3 streams + CLAM_SB on a 100 000 x 384 bag)
Per leg: W mean / max, sclk MHz mean / min, achieved TFLOP/s (HIP events), and for the bare loops the in-kernel clock (s_memtime / s_memrealtime).
    python tools/power_legs.py [--seconds 5] [--skip 1.5] [--out profiles/r06_power.json]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=5.0)
    ap.add_argument("--skip", type=float, default=1.5)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_power.json"))
    ap.add_argument("--legs", default="idle,mfma16,mfma32,mlp16,qkv_attn,step,idle2")
    args = ap.parse_args()

    import numpy as np
    import torch

    from telemetry import PowerSampler, pci_bus_id_of
    from hipt_abmil_atec23_amd import _native as N, functional as Fn, synth
    from hipt_abmil_atec23_amd.vision_transformer import vit_small

    dev = torch.device("cuda:0")
    pci = pci_bus_id_of(0)
    ps = PowerSampler(pci, period_s=0.005)
    res = {"_meta": {"pci": pci, "hwmon": ps.dir, "power_cap_w": ps.cap_w, "seconds_per_leg": args.seconds, "skipped_head_s": args.skip,
                     "device": torch.cuda.get_device_name(0),
                     "how": "tools/power_legs.py: one process, sampler thread on the board's hwmon node (power1_input = socket power, freq1_input = sclk) "
                            "every 5 ms while the main thread keeps the leg's kernel(s) queued back to back"}}
    if not ps.available:
        res["_meta"]["error"] = "no readable hwmon node for this device"
        json.dump(res, open(args.out, "w"), indent=1)
        print(json.dumps(res))
        return
    ps.start()

    def hold(name, enqueue, flop_per_call=None, extra=None, sync_every=4):
        """keep `enqueue` queued for args.seconds (a few calls ahead of the GPU), HIP events around the whole leg"""
        enqueue()
        torch.cuda.synchronize()
        lo = ps.mark()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        n = 0
        while time.perf_counter() - t0 < args.seconds:
            for _ in range(sync_every):
                enqueue()
                n += 1
            ev = torch.cuda.Event()
            ev.record()
            while not ev.query() and time.perf_counter() - t0 < args.seconds + 5:
                time.sleep(0.001)
        b.record()
        torch.cuda.synchronize()
        hi = ps.mark()
        ms = a.elapsed_time(b)
        r = ps.summary(lo, hi, skip_s=args.skip)
        r.update({"calls": n, "ms_per_call": ms / n})
        if flop_per_call:
            r["tflops"] = flop_per_call * n / (ms * 1e-3) / 1e12
        if extra:
            r.update(extra())
        res[name] = r
        print(name, json.dumps(r), file=sys.stderr, flush=True)

    legs = args.legs.split(",")
    if "idle" in legs:
        torch.cuda.synchronize()
        lo = ps.mark()
        time.sleep(min(args.seconds, 3.0))
        res["idle"] = ps.summary(lo, ps.mark())

    # ---- bare MFMA loops ----
    so = os.path.join(ROOT, "tools", "probe_bin", "libpower_probe.so")
    if ("mfma16" in legs or "mfma32" in legs):
        if not os.path.exists(so):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(ROOT, "tools", "power_probe.hip"), "-o", so])
        pp = C.CDLL(so)
        pp.pp_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        pp.pp_flop_per_iter.restype, pp.pp_flop_per_iter.argtypes = C.c_double, [C.c_int]
        ncu = torch.cuda.get_device_properties(0).multi_processor_count
        outb = torch.zeros(2 * 4 * ncu, dtype=torch.int64, device=dev)
        iters = 400000  # 51 M matrix-pipe cycles per launch: ~25 ms
        for shape in (16, 32):
            if f"mfma{shape}" not in legs:
                continue

            def clock():
                h = outb.cpu().numpy().reshape(-1, 2).astype(np.float64)
                ghz = h[:, 0] / h[:, 1] * 0.1
                return {"in_kernel_ghz_median": float(np.median(ghz)), "in_kernel_ghz_min": float(ghz.min()), "in_kernel_ghz_max": float(ghz.max()),
                        "cycles_per_iteration": float(np.median(h[:, 0]) / iters), "pipe_cycles_per_iteration": 128 if shape == 32 else 256}
            hold(f"mfma_{shape}x{shape}x{512 // shape}_bare", lambda: pp.pp_launch(shape, ncu, iters, outb.data_ptr(), torch.cuda.current_stream().cuda_stream),
                 flop_per_call=ncu * 4 * iters * pp.pp_flop_per_iter(shape), extra=clock)

    # ---- the two dominant kernels alone, 2 048 patches per launch ----
    if "mlp16" in legs or "qkv_attn" in legs:
        m = vit_small(patch_size=16, num_classes=0)
        m.load_state_dict(synth.make_state_dict(synth.vit_param_specs("vit256"), 256))
        m = m.eval().to(dev).set_compute_dtype("bf16")
        pk = m._tokens(synth.hash_uniform_torch((1, 3, 256, 256), 2, device=dev))[0]
        nseq = 2048
        M = nseq * 257
        xn = torch.randn(M, 384, device=dev).bfloat16()
        att = torch.empty_like(xn)
        ws = Fn.workspace(dev, N.lib().hipt_vit_workspace_bytes(pk.ref, nseq))
        st = N.stream_ptr(dev)
        if "qkv_attn" in legs:
            fl = 2 * M * 1152 * 384 + 4 * nseq * 6 * 257 * 257 * 64
            hold("qkv_attn_alone", lambda: N.call("hipt_vit_attention_unit", pk.ref, 3, N.ptr(xn), nseq, N.ptr(att), 1, N.ptr(ws), ws.numel(), st), flop_per_call=fl)
        if "mlp16" in legs:
            N.call("hipt_vit_attention_unit", pk.ref, 3, N.ptr(xn), nseq, N.ptr(att), 1, N.ptr(ws), ws.numel(), st)
            x = torch.randn(M, 384, device=dev)
            x0 = x.clone()
            xo = torch.empty_like(xn)
            cnt = torch.zeros(64, dtype=torch.int32, device=dev)
            k = [0]

            def mlp():
                if k[0] % 16 == 0:
                    x.copy_(x0)  # (the residual stream is updated in place: back to O(1) values every 16 launches)
                k[0] += 1
                N.call("hipt_vit_mlp_unit", pk.ref, 3, N.ptr(x), N.ptr(att), nseq, N.ptr(xo), N.ptr(cnt), cnt.numel() * 4, st)
            hold("mlp16_alone", mlp, flop_per_call=M * (4 * 384 * 1536 + 2 * 384 * 384))
            del x, x0, xo
        del xn, att

    # ---- the full bench step ----
    if "step" in legs:
        import bench as B
        model, clam, _ = B.build_models(dev, "bf16")
        model.streams = 3
        R = 24
        region = synth.hash_uniform_torch((R, 3, 4096, 4096), 3, device=dev)
        bags = [synth.hash_uniform_torch((100000, 384), 40 + i, device=dev).to(torch.bfloat16) for i in range(5)]
        k2 = [0]

        def step():
            model(region)
            with torch.no_grad():
                clam(bags[k2[0] % 5])
            k2[0] += 1
        hold("full_step_24_regions", step, flop_per_call=R * B.FLOP_PER_REGION, sync_every=2)
        res["full_step_24_regions"]["regions_per_s"] = R / (res["full_step_24_regions"]["ms_per_call"] * 1e-3)
    if "idle2" in legs:
        torch.cuda.synchronize()
        time.sleep(1.0)
        lo = ps.mark()
        time.sleep(2.0)
        res["idle_after"] = ps.summary(lo, ps.mark())
    ps.stop()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
