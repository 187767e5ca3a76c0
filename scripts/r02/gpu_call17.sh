#!/bin/bash
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call17.log; : > $L
timeout 300 python scripts/r02/check_variants.py --lattice 8,8,8,8 --time 0 --variants 6 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 16,16,16,32 --time 0 --variants 6 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 32,32,32,64 --variants 6 --nts 5 --reps 300 >> $L 2>&1
bash scripts/r02/pmc_traffic.sh pmc_v6_r12 --set gauge_recon=12 --set dslash_variant=6 >> $L 2>&1
bash scripts/r02/pmc_traffic.sh pmc_v6_r18 --set gauge_recon=18 --set dslash_variant=6 >> $L 2>&1
grep -E "VARIANTS|^time|^PMC" $L
