set -e
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
CMD="python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --slides 0 --streams 1 --profile-steps 0"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU -d $REPO/gpurun_out/pmc_lds_mlp32 -o pmc -- $CMD > $REPO/gpurun_out/pmc_lds_mlp32.log 2>&1 || echo "pass 1 failed"
export HIPT_MLP_CO=1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_INSTS_VALU -d $REPO/gpurun_out/pmc_lds_mlpco -o pmc -- $CMD > $REPO/gpurun_out/pmc_lds_mlpco.log 2>&1 || echo "pass 2 failed"
cd $REPO
find gpurun_out -name "*.db" -delete 2>/dev/null || true
ls gpurun_out/pmc_lds_mlp32 gpurun_out/pmc_lds_mlpco
