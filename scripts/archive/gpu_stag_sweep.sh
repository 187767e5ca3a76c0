#!/bin/bash
cd "$(dirname "$0")/../.."
for cfg in "16 1" "32 1" "48 1" "96 1" "24 3" "48 3" "48 6" "96 6" "32 2" "16 2" "96 3" "144 9" "144 3"; do
  set -- $cfg
  timeout 120 python scripts/dslash_probe.py --kind Staggered --lattice 48,48,48,96 --reps 60 --warm 10 --set xcd_nsub=$1 --set xcd_ysplit=$2 2>&1 | tail -1
done
for cfg in "8 1" "16 1" "32 1" "16 2" "16 4" "32 4" "32 2" "64 4"; do
  set -- $cfg
  timeout 120 python scripts/dslash_probe.py --kind Staggered --lattice 32,32,32,32 --reps 100 --warm 10 --set xcd_nsub=$1 --set xcd_ysplit=$2 2>&1 | tail -1
done
