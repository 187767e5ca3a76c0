#!/bin/bash
# variant 8 (both hops of a direction in flight, no branches in the body) against variant 1: correctness, Dslash, CG window, mixed CG
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "dirsplit_variant" 2>&1 | tail -3
for v in 1 8 1 8; do
  for recon in 12 18; do
    echo -n "variant $v recon $recon: "; python scripts/dslash_probe.py --reps 200 --warm 20 --cg 200 --set dslash_variant=$v --set gauge_recon=$recon 2>&1 | tail -1 | cut -c1-200
  done
done
for v in 1 8 1 8; do echo -n "mixed variant $v: "; LQCD_SET="dslash_variant=$v" python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 2>&1 | tail -1; done
