#!/bin/bash
set -x
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call3.log
: > $L
timeout 300 python scripts/r02/check_variants.py --lattice 8,8,8,8 --time 0 --variants 6 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 16,16,16,32 --time 0 --variants 6 >> $L 2>&1
timeout 600 python scripts/r02/check_variants.py --lattice 32,32,32,64 --variants 6 --nts 0,4,5 >> $L 2>&1
for v in 1 6; do for recon in 18 12; do
python scripts/dslash_probe.py --reps 200 --warm 20 --cg 100 --set gauge_recon=$recon --set dslash_variant=$v --set nt_store=1 >> $L 2>&1
done; done
cat $L
