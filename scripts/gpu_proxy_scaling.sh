#!/bin/bash
# one-die proxy of the strong-scaling curve: the local volumes of 32^3x64 on N = 1, 2, 4, 8 GPUs (PE grids of pegrid.choose_pe_grid), the partitioned directions exchanged
# with this rank itself through the peer-mapped backend (selfcomm 2) or RCCL (selfcomm 1); schedule chosen by the tuner.   usage: gpu_proxy_scaling.sh [selfcomm]
cd "$(dirname "$0")/.."
sc=${1:-2}
run() { LQCD_FORCE_PARTITION=$1 timeout 200 python scripts/dslash_probe.py --lattice $2 --selfcomm $sc --reps 200 --warm 20 --cg 400 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/$3 /"; echo; }
timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,64 --reps 200 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=1 (1,1,1,1) /"; echo
run 8 32,32,32,32 "N=2 (1,1,1,2)"
run 12 32,32,16,32 "N=4 (1,1,2,2)"
run 12 32,32,16,16 "N=8 (1,1,2,4)"
run 14 32,16,16,32 "N=8 (1,2,2,2)"
