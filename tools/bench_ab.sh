#!/bin/bash
# Same-box A/B of whole library builds through bench.py itself (the probe's back-to-back launches sit in a power transient that the
# pipeline does not): tools/bench_ab.sh <rounds> <lib suffix> ... ("-" = the shipped library; NAME=VALUE = the shipped library under that environment switch, e.g. HIPT_NO_PROJ_FOLD=1), interleaved; prints regions/s and the
# per-launch time of the fused MLP / fused attention kernels.
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    lib=hipt_abmil_atec23_amd/libhipt_abmil${v:+_$v}.so
    [ "$v" = "-" ] && lib=hipt_abmil_atec23_amd/libhipt_abmil.so
    ev=""
    case "$v" in *=*) ev="$v"; lib=hipt_abmil_atec23_amd/libhipt_abmil.so;; esac   # NAME=VALUE: the shipped library under that environment
    env $ev HIPT_AMD_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-extras --slides 0 > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$v FAILED"; tail -3 gpurun_out/ab_tmp.err; continue; }
    python - "$v" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
k = d.get("kernels", d.get("per_kernel", {}))
au = d['roofline_attention_unit']
print(f"{sys.argv[1]:>6}: {d['value']:.1f} regions/s  step {d['ms_per_step']:.2f} ms  mlp {d['roofline']['avg_launch_us']:.1f} us ({d['roofline']['frac']:.4f})  attn unit {au['frac']:.4f} ({au['us_per_launch_set']:.0f} us, [CLS] side {au.get('cls_rows_us_per_block', 0):.1f} us)  abmil {d['roofline_abmil']['frac']:.4f}  {d['selfcheck']}")
PY
  done
done
