#!/bin/bash
# variant 9 (branch-free body, loads in three groups, 3 waves/SIMD) against variant 1
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "dirsplit_variant or random_configuration" 2>&1 | tail -3
for v in 1 9 1 9; do
  for recon in 12 18; do
    echo -n "variant $v recon $recon: "; python scripts/dslash_probe.py --reps 200 --warm 20 --cg 200 --set dslash_variant=$v --set gauge_recon=$recon 2>&1 | tail -2 | tr '\n' ' ' | sed 's/dslash Wilson L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-200; echo
  done
done
for v in 1 9 1 9; do echo -n "mixed variant $v: "; LQCD_SET="dslash_variant=$v" python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 2>&1 | tail -1; done
