#!/bin/bash
# direction rotation across SIMDs: A = rotate (default build), B = wave w always takes direction w
cd "$(dirname "$0")/../.."
R=$(pwd)
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dirsplit_variant or wilson_dslash or clover" 2>&1 | tail -2
for v in A B A B; do
  if [ $v = B ]; then export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_b.so; else unset LQCD_HIP_LIB; fi
  for recon in 12 18; do
    echo -n "$v recon $recon: "; python scripts/dslash_probe.py --reps 200 --warm 20 --cg 200 --set gauge_recon=$recon 2>&1 | tail -2 | tr '\n' ' ' | sed 's/dslash Wilson L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-200; echo
  done
done
