#!/bin/bash
# Same-box A/B of library variants under any tool: tools/ab_tool.sh <rounds> "<python tool + args>" <lib suffix> ... ("-" = the shipped library), interleaved.
R=$1; T=$2; shift; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    lib=hipt_abmil_atec23_amd/libhipt_abmil_$v.so
    [ "$v" = "-" ] && lib=hipt_abmil_atec23_amd/libhipt_abmil.so
    echo "== $v"; HIPT_AMD_LIB=$PWD/$lib python $T 2>&1 | grep -v amdgpu
  done
done
