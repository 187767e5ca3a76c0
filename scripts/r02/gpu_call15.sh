#!/bin/bash
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call15.log; : > $L
export LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip_ablate.so
for recon in 18 12; do for dbg in 0 1048576 2097152 3145728; do
python scripts/dslash_probe.py --reps 300 --warm 30 --set gauge_recon=$recon --set dbg=$dbg >> $L 2>&1
done; done
grep "^dslash" $L
