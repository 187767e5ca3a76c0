#!/bin/bash
# re-tune the workgroup -> lattice map for the 12-real kernel (the defaults were tuned on the 18-real one)
cd "$(dirname "$0")/.."
for remap in 2 1; do for nsub in 8 16 32; do for ys in 2 4 8; do
timeout 60 python scripts/dslash_probe.py --reps 200 --warm 20 --set xcd_remap=$remap --set xcd_nsub=$nsub --set xcd_ysplit=$ys 2>&1 | grep "^dslash" | sed "s/^/remap=$remap nsub=$nsub ysplit=$ys /" | cut -c1-140
done; done; done
