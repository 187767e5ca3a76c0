#!/bin/bash
# which of the instruction-count reductions costs time?  A all on | B previous commit | C no sign skip | D no SET | E neither (row2 + mv only) | F mv only
cd "$(dirname "$0")/../.."
R=$(pwd)
for round in 1 2; do
for v in A B C D E F; do
  case $v in A) unset LQCD_HIP_LIB;; *) export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_$(echo $v | tr A-Z a-z).so;; esac
  echo -n "$v recon 12: "; python scripts/dslash_probe.py --reps 200 --warm 20 --set gauge_recon=12 2>&1 | tail -1 | sed 's/dslash Wilson L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-100
done; done
