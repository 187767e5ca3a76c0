#!/bin/bash
# fabric traffic (PMC) of the staggered split kernel, 12- and 18-real links, 48^3x96 and 32^3x64
cd "$(dirname "$0")/../.."
for L in 32,32,32,64 48,48,48,96; do
  for recon in 12 18; do
    bash scripts/r02/pmc_traffic.sh stag_${L//,/x}_r$recon --lattice $L --kind Staggered --set gauge_recon=$recon 2>&1 | grep "^PMC\|failed"
  done
done
