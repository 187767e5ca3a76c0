#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 --cg 100 "$@" 2>&1 | grep -E "^dslash|^cg" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
LQCD_FORCE_PARTITION=14 run n8_local_part_yzt --lattice 32,16,16,32 --selfcomm 1
LQCD_FORCE_PARTITION=12 run n4_local_part_zt --lattice 32,32,16,32 --selfcomm 1
LQCD_FORCE_PARTITION=8 run n2_local_part_t --lattice 32,32,32,32 --selfcomm 1
