#!/bin/bash
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call19.log; : > $L
timeout 300 python scripts/r02/check_variants.py --lattice 8,8,8,8 --time 0 --variants 7 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 16,16,16,32 --time 0 --variants 7 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 32,32,32,64 --variants 5,7 --nts 5 --reps 300 >> $L 2>&1
bash scripts/r02/pmc_traffic.sh pmc_v7_r12 --set gauge_recon=12 --set dslash_variant=7 >> $L 2>&1
python scripts/dslash_probe.py --reps 100 --warm 10 --cg 200 --set dslash_variant=7 >> $L 2>&1
python scripts/dslash_probe.py --reps 100 --warm 10 --cg 200 --set dslash_variant=1 >> $L 2>&1
grep -E "VARIANTS|^time|^PMC|^cg|^dslash" $L
