#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 300 python -m pytest tests -m gpu -q -x -k "staggered or Staggered or partitioned or rccl or multishift or force" 2>&1 | tail -4
for L in 32,32,32,32 48,48,48,96 16,16,16,32 8,8,8,8; do
  for v in 0 1; do
    timeout 120 python scripts/dslash_probe.py --kind Staggered --lattice $L --reps 100 --warm 10 --cg 50 --set dslash_variant=$v 2>&1 | tail -2 | tr '\n' ' '; echo
  done
done
