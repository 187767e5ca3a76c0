#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_gpu_recon12.py -m gpu -q -x -s 2>&1 | tail -5
for r in 18 12 18 12; do
  timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 --cg 50 --set gauge_recon=$r 2>&1 | tail -2 | tr '\n' ' '; echo
done
