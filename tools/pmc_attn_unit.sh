#!/bin/bash
# PMC passes over the attention unit alone (tools/attn_unit_bench.py: fused kernel, 2048 patches): wave-cycle breakdown, MFMA busy, VALU / LDS
# instruction counts, LDS bank conflicts, HBM traffic.  Run from the repo root through gpurun; outputs under gpurun_out/pmc_attn_<pass>/.
set -e
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
CMD="python3 $REPO/tools/attn_unit_bench.py 2048 ${1:-1}"
T=${2:-f}
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES -d $REPO/gpurun_out/pmc_attn_${T}_a -o pmc -- $CMD > $REPO/gpurun_out/pmc_attn_${T}_a.log 2>&1 || echo "pass a failed"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM -d $REPO/gpurun_out/pmc_attn_${T}_b -o pmc -- $CMD > $REPO/gpurun_out/pmc_attn_${T}_b.log 2>&1 || echo "pass b failed"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc_attn_${T}_c -o pmc -- $CMD > $REPO/gpurun_out/pmc_attn_${T}_c.log 2>&1 || echo "pass c failed"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmc_attn_${T}_d -o pmc -- $CMD > $REPO/gpurun_out/pmc_attn_${T}_d.log 2>&1 || echo "pass d failed"
cd $REPO
find gpurun_out -name "*.db" -delete 2>/dev/null || true
python3 - <<PY
import csv, glob, collections
for ps in "abcd":
    for f in glob.glob("gpurun_out/pmc_attn_${T}_%s/**/*counter_collection.csv" % ps, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "qkv_attn" in n or "attn64" in n or "seqgemm_pipe" in n:
                agg[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            print(k, {c: sorted(v)[len(v) // 2] for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
