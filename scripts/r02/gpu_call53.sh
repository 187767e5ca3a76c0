#!/bin/bash
# map sweep for the all-18-reals Wilson kernel at 32^3x64 (is the default 16/4 still the best point for it?)
cd "$(dirname "$0")/../.."
for rep in 1 2; do
for nsub in 8 16 32; do
  for ys in 1 2 4 8; do
    echo -n "nsub $nsub ysplit $ys: "; python scripts/dslash_probe.py --reps 200 --warm 20 --set gauge_recon=18 --set xcd_nsub=$nsub --set xcd_ysplit=$ys 2>&1 | tail -1 | sed 's/dslash Wilson L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-40
  done
done; done
