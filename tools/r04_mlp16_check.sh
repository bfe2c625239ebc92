P=tools/probe_bin
for b in mlp_probe32 mlp_probe16; do
for a in "check 514" "check 100 7"; do $P/$b $a 2>&1 | tail -1; done
PROBE_IMG=1 $P/$b check 528 16 2>&1 | tail -1
PROBE_IMG=3 $P/$b check 2064 32 2>&1 | tail -1
PROBE_IMG=3 $P/$b check 4128 128 2>&1 | tail -1
PROBE_IMG=3 $P/$b check 65792 2>&1 | tail -1
done
