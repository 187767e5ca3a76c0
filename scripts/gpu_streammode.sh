#!/bin/bash
# proxy of the 8-GPU local volume on one GPU (self-partition y,z,t + world-size-1 RCCL): halo_stream_mode 0 vs 1
cd "$(dirname "$0")/.."
export LQCD_FORCE_PARTITION=14
for rep in 1 2; do
for m in 0 1; do
timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 --set halo_stream_mode=$m 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/mode=$m /"; echo
done
done
for m in 0 1; do
LQCD_SET="halo_stream_mode=$m" timeout 200 python scripts/mixed_probe.py 32,16,16,32 Wilson 1e-16 2>&1 | grep -E "fp64|mixed" | sed "s/^/mode=$m /"
done
unset LQCD_FORCE_PARTITION
LQCD_FORCE_PARTITION=15 LQCD_SET="halo_stream_mode=1" timeout 200 python scripts/mixed_probe.py 16,16,16,16 Staggered 1e-16 2>&1 | grep -E "fp64|mixed|Error|error" | sed "s/^/stag mask15 mode=1 /"
