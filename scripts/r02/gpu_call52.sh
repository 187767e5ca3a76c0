#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_md.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
for L in 48,48,48,96 48,24,24,48 32,32,32,64; do for k in Staggered Wilson; do
  echo -n "$k $L: "; python scripts/dslash_probe.py --lattice $L --kind $k --reps 60 --warm 5 --cg 60 2>&1 | tail -2 | tr '\n' ' ' | sed 's/dslash [A-Za-z]* L=([0-9, ]*) set=\[[^]]*\] //' | cut -c1-150; echo
done; done
