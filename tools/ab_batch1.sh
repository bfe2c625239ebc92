#!/bin/bash
# Same-box A/B of library builds at one region per call: tools/ab_batch1.sh <rounds> <lib suffix> ... ("-" = the shipped library)
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    lib=hipt_abmil_atec23_amd/libhipt_abmil_$v.so
    [ "$v" = "-" ] && lib=hipt_abmil_atec23_amd/libhipt_abmil.so
    echo -n "$v: "; HIPT_AMD_LIB=$PWD/$lib python tools/batch1_bench.py 2 | tail -1
  done
done
