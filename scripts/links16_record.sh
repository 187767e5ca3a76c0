#!/bin/bash
# the measurements behind profiles/r06_links16_reliable.log (one gpurun call)
cd "$(dirname "$0")/.."
echo "# int16 links in the fp32 site-pair inner operator (mixed_links16) and reliable updates (bicg_reliable); 32^3x64, hot configuration, one MI355X"
python scripts/links16_probe.py 2>&1
export LQCD_MIXED_TRACE=1
for k in 0.141139 0.19 0.22; do
  for l16 in 0 1; do for rel in 0 1; do
    echo "== e-o BiCGStab, kappa $k, mixed, links16 $l16, reliable $rel"
    python scripts/bicg_probe.py 2 --L 32,32,32,64 --reps 2 --kappa $k --set bicg_mixed=1 --set mixed_links16=$l16 --set bicg_reliable=$rel 2>&1 | tail -4 | cut -c1-220
  done; done
  echo "== e-o BiCGStab, kappa $k, fp64"
  python scripts/bicg_probe.py 2 --L 32,32,32,64 --reps 2 --kappa $k 2>&1 | tail -1
done
unset LQCD_MIXED_TRACE
for m in 0 1; do python scripts/md_probe.py $m 2>&1 | grep "MD step"; done
