#!/bin/bash
# CG iterations/s at the local volumes of N = 1, 2, 4, 8 GPUs on ONE GPU (self-partition + world-size-1 RCCL: all halo machinery active)
cd "$(dirname "$0")/.."
timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,64 --reps 200 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=1 /"; echo
LQCD_FORCE_PARTITION=8 timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,32 --selfcomm 1 --reps 200 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=2 (1,1,1,2) /"; echo
LQCD_FORCE_PARTITION=12 timeout 200 python scripts/dslash_probe.py --lattice 32,32,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=4 (1,1,2,2) /"; echo
LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 (1,2,2,2) /"; echo
