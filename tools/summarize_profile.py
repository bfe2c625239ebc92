#!/usr/bin/env python3
"""Turn rocprofv3 outputs (gpurun_out/) into the committed summaries under profiles/.

    python tools/summarize_profile.py r01

Reads   gpurun_out/prof_<tag>/**/_kernel_stats.csv          (--kernel-trace --stats)
        gpurun_out/pmc_{fetch,write,sq,grbm,lds,wait}_<tag>/**/_counter_collection.csv   (separate --pmc passes)
Writes  profiles/<tag>_kernel_stats.csv   our kernels only, per-kernel calls / total / average duration
        profiles/<tag>_pmc.csv            per-kernel medians of the counters
        profiles/<tag>_traffic.json       per-launch HBM bytes of the hot kernels, corrected as
                                          MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE (KB) x 1024 x 2
                                          (gfx950 counts 128-B requests as 64 B), WRITE_SIZE (KB) x 1024
"""
import collections
import csv
import glob
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    m = re.search(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I(.*?)EEv", n)
    if m:
        n = f"{m.group(1)}<{m.group(2)}>"
    return n.split("(")[0][:80]


def ours(name):
    return any(k in name for k in ("seqgemm", "mlp_kernel", "mlp_pipe_kernel", "mlp32_kernel", "mlp16_kernel", "mlp16_pack", "qkv_attn", "mlp_co_kernel", "mlp_ws_kernel", "mlp32_pack", "mlp_ws_pack", "embed32", "gemm_kernel", "attn_kernel", "attn64_kernel", "attn_cls_kernel", "gather_cls", "u8_norm", "ln_kernel", "abmil", "mlp_pack",
                                   "cls_init", "f32_to_bf16", "gate_kernel", "gemm_small", "lngemm_small", "clam_mb_pool", "pool_kernel", "add_bf16", "clam_train", "topk_rows", "attn_cls_probs"))


stats = glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = [r for r in csv.DictReader(open(stats[0])) if ours(r["Name"])]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct_of_our_kernels"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], f"{float(r['TotalDurationNs']) / 1e3:.1f}", f"{float(r['AverageNs']) / 1e3:.2f}",
                        f"{float(r['MinNs']) / 1e3:.2f}", f"{float(r['MaxNs']) / 1e3:.2f}", f"{100 * float(r['TotalDurationNs']) / tot:.2f}"])
    print("wrote", f"profiles/{tag}_kernel_stats.csv", len(rows), "kernels")

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write", "sq", "grbm", "lds", "wait"):
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{kind}_{tag}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


# the big ViT-256 launches dominate each kernel's list; the ViT-4K launches (257 rows) are tiny -> use the max-side median
def med_big(v):
    v = sorted(v)
    return v[(len(v) * 3) // 4] if len(v) > 3 else v[-1]


if agg:
    counters = sorted({c for k in agg.values() for c in k})
    with open(os.path.join(OUT, f"{tag}_pmc.csv"), "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches"] + counters)
        for k, d in sorted(agg.items()):
            w.writerow([k, max(len(x) for x in d.values())] + [f"{med_big(d[c]):.6g}" if c in d else "" for c in counters])
    traffic = {}
    for k, d in agg.items():
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            traffic[k] = {"fetch_bytes": med_big(d["FETCH_SIZE"]) * 1024 * 2, "write_bytes": med_big(d["WRITE_SIZE"]) * 1024,
                          "note": "per launch (upper-quartile launch = ViT-256 shape); FETCH_SIZE doubled per MI355X_MICROARCH.md"}
            traffic[k]["hbm_bytes"] = traffic[k]["fetch_bytes"] + traffic[k]["write_bytes"]
    # the commit whose library these counters were taken on (bench.py quotes it as roofline.traffic_source: the GPU box has no .git)
    try:
        import subprocess
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:
        head = ""
    traffic["_meta"] = {"tag": tag, "measured_at_commit": head or None,
                        "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --steps 3 --warmup 1 --streams 1` (tools/run_profile.sh)"}
    json.dump(traffic, open(os.path.join(OUT, f"{tag}_traffic.json"), "w"), indent=1)
    print("wrote", f"profiles/{tag}_pmc.csv, profiles/{tag}_traffic.json", len(traffic), "kernels with traffic")
