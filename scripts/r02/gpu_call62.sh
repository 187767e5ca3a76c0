#!/bin/bash
# branch-free hops on unpartitioned lattices (PART = false): A = on (fp32 under the 96-VGPR cap: spills 20), B = off (previous), C = on with the fp32 cap at 4 workgroups/CU (106 VGPRs)
cd "$(dirname "$0")/../.."
R=$(pwd)
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "wilson_dslash or dirsplit_variant or staggered_dslash or recon" 2>&1 | tail -2
for v in A B C A B C; do
  case $v in A) unset LQCD_HIP_LIB;; B) export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_b.so;; C) export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_c.so;; esac
  for recon in 12 18; do
    echo -n "$v recon $recon: "; python scripts/dslash_probe.py --reps 200 --warm 20 --cg 200 --set gauge_recon=$recon 2>&1 | tail -2 | tr '\n' ' ' | sed 's/dslash Wilson L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-170; echo
  done
  echo -n "$v mixed: "; python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 2>&1 | tail -1
done
