#!/bin/bash
# Round 4, same-box A/B of the fused MLP: tools/probe_bin/mlp_probe32 (csrc/mlp32.hip; mask 64 = without the L2 prefetch of the next
# tile's rows) against mlp_probe32r3 (round 3's kernel), both built WITHOUT stamps; then the stamped build of the new kernel (tile 8).
P=tools/probe_bin
O=gpurun_out/r04_mlp_ab.log
mkdir -p gpurun_out
: > $O
run() { echo "== $*" >> $O; timeout -k 10 120 "$@" >> $O 2>&1; echo "rc=$?" >> $O; }
run $P/mlp_probe32 check 514
run $P/mlp_probe32 check 100 7
PROBE_IMG=1 run $P/mlp_probe32 check 528 16
PROBE_IMG=3 run $P/mlp_probe32 check 2064 32
PROBE_IMG=3 run $P/mlp_probe32 check 4128 128
grep -q "rc=1" $O && { echo "CHECK FAILED"; tail -30 $O; exit 1; }
for i in 1 2 3; do
PROBE_IMG=1 run $P/mlp_probe32 8 0 64
PROBE_IMG=1 run $P/mlp_probe32r3 8 0
done
PROBE_IMG=1 run $P/mlp_probe32 1 0 64
PROBE_IMG=1 run $P/mlp_probe32r3 1 0
PROBE_IMG=1 HIPT_SEQGEMM_STAMPS=1 run $P/mlp_probe32_st8 8 0 64
grep -v "^\[mlp32 dbg=\(8\|15\|2\) " $O | grep -v "dbg=\(2\|8\|15\) "
