#!/bin/bash
# round 3, GPU call L: fp32 inner kernel with both hops in flight (separate build) vs default, mixed CG A/B on one box
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_l; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for lib in "" _both32; do
  echo "== lib$lib"
  LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip$lib.so timeout 300 python scripts/r03/mixed_ab.py
  LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip$lib.so timeout 300 python scripts/r03/mixed_ab.py dslash_pipe=0
done; done 2>&1 | tee $O/mixed_ab.log
