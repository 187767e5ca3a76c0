#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 300 python -m pytest tests/test_gpu_recon12.py -m gpu -q -x 2>&1 | tail -3
for r in 18 12; do
  timeout 120 python scripts/dslash_probe.py --kind Staggered --lattice 48,48,48,96 --reps 100 --warm 10 --cg 50 --set gauge_recon=$r 2>&1 | tail -2 | tr '\n' ' '; echo
  timeout 120 python scripts/dslash_probe.py --kind Staggered --lattice 32,32,32,32 --reps 100 --warm 10 --cg 50 --set gauge_recon=$r 2>&1 | tail -2 | tr '\n' ' '; echo
done
