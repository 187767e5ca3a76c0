#!/bin/bash
cd "$(dirname "$0")/../.."
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 --cg 100 "$@" 2>&1 | grep -E "^dslash|^cg" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
echo "--- local volume of N=8 (1,2,2,2): 32x16x16x32"
run n8_local_nopart --lattice 32,16,16,32
LQCD_FORCE_PARTITION=14 run n8_local_part_yzt --lattice 32,16,16,32 --selfcomm 1
echo "--- local volume of N=4 (1,1,2,2): 32x32x16x32"
run n4_local_nopart --lattice 32,32,16,32
LQCD_FORCE_PARTITION=12 run n4_local_part_zt --lattice 32,32,16,32 --selfcomm 1
echo "--- local volume of N=2 (1,1,1,2): 32x32x32x32"
run n2_local_nopart --lattice 32,32,32,32
LQCD_FORCE_PARTITION=8 run n2_local_part_t --lattice 32,32,32,32 --selfcomm 1
echo "--- unfused CG on the full lattice (what a partitioned rank runs)"
run full_cgfused1 --set cg_fused=1
