#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/run_profile.sh r01
# 1) --kernel-trace --stats of the default bench command (per-kernel durations),
# 2) separate --pmc passes (FETCH_SIZE / WRITE_SIZE do not fit one pass; never combined with trace domains other
#    than --kernel-trace) for the HBM traffic of the hot kernels.
# Outputs land under gpurun_out/; tools/summarize_profile.py <tag> turns them into profiles/<tag>_*.
set -e
TAG=${1:-r01}
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
# one stream: with the default three, launches of different streams share the CUs and a kernel's duration in the
# trace is no longer its own (the roofline leg inside bench.py measures on one stream for the same reason)
CMD="python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --slides 0 --streams 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o prof -- $CMD > $REPO/gpurun_out/prof_$TAG.log 2>&1
echo "kernel trace done"
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $REPO/gpurun_out/pmc_fetch_$TAG -o pmc -- $CMD > $REPO/gpurun_out/pmc_fetch_$TAG.log 2>&1
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $REPO/gpurun_out/pmc_write_$TAG -o pmc -- $CMD > $REPO/gpurun_out/pmc_write_$TAG.log 2>&1
echo "WRITE_SIZE pass done"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES -d $REPO/gpurun_out/pmc_sq_$TAG -o pmc -- $CMD > $REPO/gpurun_out/pmc_sq_$TAG.log 2>&1 || echo "SQ pass failed (counter names)"
echo "SQ pass done"
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d $REPO/gpurun_out/pmc_grbm_$TAG -o pmc -- $CMD > $REPO/gpurun_out/pmc_grbm_$TAG.log 2>&1
echo "GRBM pass done"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS -d $REPO/gpurun_out/pmc_lds_$TAG -o pmc -- $CMD > $REPO/gpurun_out/pmc_lds_$TAG.log 2>&1 || echo "LDS pass failed (counter names)"
echo "LDS pass done"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $REPO/gpurun_out/pmc_wait_$TAG -o pmc -- $CMD > $REPO/gpurun_out/pmc_wait_$TAG.log 2>&1 || echo "wait pass failed (counter names)"
echo "wait pass done"
cd $REPO
# keep only the CSVs the summariser reads (the merge back is capped at 64 MiB)
find gpurun_out -name "*.db" -delete 2>/dev/null || true
du -sh gpurun_out | tail -1
