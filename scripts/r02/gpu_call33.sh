#!/bin/bash
# staggered split kernel: hop by hop (stag_both = 0) against both hops in flight (1), same box
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -k "staggered_both or random_configuration" 2>&1 | tail -3
for L in 32,32,32,64 48,48,48,96; do
for b in 0 1 0 1; do
  for recon in 12 18; do
    echo -n "L $L stag_both $b recon $recon: "; python scripts/dslash_probe.py --lattice $L --kind Staggered --reps 100 --warm 10 --cg 100 --set stag_both=$b --set gauge_recon=$recon 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-260; echo
  done
done; done
