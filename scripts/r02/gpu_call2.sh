#!/bin/bash
set -x
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call2.log
: > $L
timeout 300 python scripts/r02/check_variants.py --lattice 8,8,8,8 --time 0 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 16,16,16,32 --time 0 >> $L 2>&1
timeout 600 python scripts/r02/check_variants.py --lattice 32,32,32,64 --nts 0,1,2,3,4,7 >> $L 2>&1
cat $L
