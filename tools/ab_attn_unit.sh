#!/bin/bash
# Same-box A/B of library variants on the attention unit alone: tools/ab_attn_unit.sh <rounds> <lib suffix> ... ("-" = the shipped library), interleaved.
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    lib=hipt_abmil_atec23_amd/libhipt_abmil_$v.so
    [ "$v" = "-" ] && lib=hipt_abmil_atec23_amd/libhipt_abmil.so
    echo -n "$v: "; HIPT_AMD_LIB=$PWD/$lib python tools/attn_unit_bench.py 2048 1 2>&1 | grep -v amdgpu | tail -1
  done
done
