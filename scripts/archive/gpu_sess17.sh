#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd)
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 --cg 100 "$@" 2>&1 | grep -E "^dslash|^cg" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
for rep in 1 2 3; do
  run aosoa_v2
  LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_soa.so run soa_v2
  run aosoa_v1 --set dslash_variant=1
  LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_soa.so run soa_v1 --set dslash_variant=1
done
run aosoa_v0 --set dslash_variant=0 --set dslash_block=64 --set lds_pad_kb=20
