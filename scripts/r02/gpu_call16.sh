#!/bin/bash
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call16.log; : > $L
timeout 300 python scripts/r02/check_variants.py --lattice 8,8,8,8 --time 0 --variants 6 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 16,16,16,32 --time 0 --variants 6 >> $L 2>&1
timeout 300 python scripts/r02/check_variants.py --lattice 32,32,32,64 --variants 6 --nts 5 --reps 300 >> $L 2>&1
echo "== occupancy-2 build of variant 6" >> $L
LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip_v6b.so timeout 300 python scripts/r02/check_variants.py --lattice 32,32,32,64 --variants 6 --nts 5 --reps 300 >> $L 2>&1
grep -E "VARIANTS|^time|^==" $L
