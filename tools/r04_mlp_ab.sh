#!/bin/bash
# Round 4, same-box A/B of the fused MLP (both probes built WITHOUT -DHIPT_DEBUG_STAMPS: the stamp branches change hipcc's
# register allocation -- 353 spills in the new kernel's stamp build against 2 in the release build -- so stamp builds are not timed).
P=tools/probe_bin
O=gpurun_out/r04_mlp_ab.log
mkdir -p gpurun_out
: > $O
run() { echo "== $*" >> $O; timeout -k 10 120 "$@" >> $O 2>&1; echo "rc=$?" >> $O; }
run $P/mlp_probe32 check 514
PROBE_IMG=3 run $P/mlp_probe32 check 2064 32
grep -q "rc=1" $O && { echo "CHECK FAILED"; tail -30 $O; exit 1; }
for i in 1 2 3; do
PROBE_IMG=1 run $P/mlp_probe32 8 0 2 8 15
PROBE_IMG=1 run $P/mlp_probe32r3 8 0 2 8 15
done
PROBE_IMG=1 run $P/mlp_probe32 1 0
PROBE_IMG=1 run $P/mlp_probe32r3 1 0
cat $O
