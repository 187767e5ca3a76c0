#!/bin/bash
# VALU diet of the direction-split kernel (first hop writes instead of adds, sign multiply skipped when every lane has +1, one-chain third row,
# no accumulator clears): A = new (default build), B = previous commit
cd "$(dirname "$0")/../.."
R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recon12.py tests/test_gpu_fuzz.py tests/test_gpu_clover.py -x -q 2>&1 | tail -3
for v in A B A B; do
  if [ $v = B ]; then export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_b.so; else unset LQCD_HIP_LIB; fi
  for recon in 12 18; do
    echo -n "$v recon $recon: "; python scripts/dslash_probe.py --reps 200 --warm 20 --cg 200 --set gauge_recon=$recon 2>&1 | tail -2 | tr '\n' ' ' | sed 's/dslash Wilson L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-200; echo
  done
done
for v in A B A B; do
  if [ $v = B ]; then export LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_b.so; else unset LQCD_HIP_LIB; fi
  echo -n "$v mixed: "; python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 2>&1 | tail -1
done
