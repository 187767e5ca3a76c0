#!/bin/bash
# fp32 update-mode r prefetch: A = none (default so far), B = before the hops (88 VGPRs, no spill since the instruction diet), C = after the hops
cd "$(dirname "$0")/../.."
R=$(pwd)
for v in A B C A B C; do
  case $v in A) unset LQCD_HIP_LIB;; B) export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_b.so;; C) export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_c.so;; esac
  echo -n "$v "; python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 2>&1 | tail -1
done
