#!/bin/bash
# PMC passes over the CLAM streaming kernel alone (tools/abmil_bench.py), one counter group per run (never with other trace domains):
#   gpurun -- 'bash tools/pmc_abmil.sh'   -> gpurun_out/pmc_abmil_<group>/ ; summary printed at the end
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $REPO/gpurun_out/pmc_abmil_$i -o pmc -- python3 $REPO/tools/abmil_bench.py > $REPO/gpurun_out/pmc_abmil_$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $REPO
python3 - <<'PY'
import csv, collections, glob
for d in sorted(glob.glob("gpurun_out/pmc_abmil_*/pmc_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "abmil32_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:36s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
