#!/bin/bash
cd "$(dirname "$0")/../.."
for pad in 0 8 16 24 32 48 64 96; do
  timeout 120 python scripts/dslash_probe.py --kind Staggered --lattice 48,48,48,96 --reps 60 --warm 10 --set lds_pad_kb=$pad 2>&1 | tail -1
done
for pad in 0 8 16 24 32 48; do
  timeout 120 python scripts/dslash_probe.py --reps 100 --warm 10 --set lds_pad_kb=$pad 2>&1 | tail -1
done
for pad in 0 16 32 64; do
  timeout 120 python scripts/dslash_probe.py --kind Staggered --lattice 32,32,32,32 --reps 100 --warm 10 --set lds_pad_kb=$pad 2>&1 | tail -1
done
