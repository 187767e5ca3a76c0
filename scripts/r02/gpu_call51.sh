#!/bin/bash
# workgroup-map sweep for the staggered kernel at 48^3x96 (864 chunks per t-slice per parity, 18 per z-plane): xcd_nsub x xcd_ysplit
cd "$(dirname "$0")/../.."
for nsub in 8 16 24 32 48 72 96 144; do
  for ys in 1 2 3 6; do
    echo -n "nsub $nsub ysplit $ys: "; python scripts/dslash_probe.py --lattice 48,48,48,96 --kind Staggered --reps 40 --warm 5 --set xcd_nsub=$nsub --set xcd_ysplit=$ys 2>&1 | tail -1 | sed 's/dslash Staggered L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-80
  done
done
