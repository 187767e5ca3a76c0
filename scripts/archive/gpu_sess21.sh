#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd)
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 300 --warm 30 "$@" 2>&1 | grep -E "^dslash|^cg" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
for rep in 1 2 3; do
run g1_v1
LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_g2.so run g2_v1
LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_g2.so run g2_v2 --set dslash_variant=2
done
LQCD_HIP_LIB=$R/latticeqcd.jl_amd/csrc/liblqcd_hip_g2.so timeout 600 python -m pytest tests -m gpu -q -x -k "oracle or fixture" 2>&1 | tail -2
