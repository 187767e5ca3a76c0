#!/bin/bash
# round 3, GPU call J: persistent kernel, sweep of the number of persistent workgroups (in-flight footprint vs memory-level parallelism)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_j; rm -rf $O; mkdir -p $O
for g in 256 320 384 448 512 576 640 768; do
  timeout 100 python scripts/dslash_probe.py --reps 200 --warm 20 --set dslash_pipe=1 --set pipe_grid=$g 2>&1 | grep ^dslash | sed "s/^/grid=$g /"
done | tee $O/grid_sweep.log
timeout 100 python scripts/dslash_probe.py --reps 200 --warm 20 --set dslash_pipe=2 2>&1 | grep ^dslash | sed "s/^/scalar /" | tee -a $O/grid_sweep.log
for g in 384 512; do
  timeout 100 python scripts/dslash_probe.py --reps 200 --warm 20 --set dslash_pipe=1 --set pipe_grid=$g --set gauge_recon=18 2>&1 | grep ^dslash | sed "s/^/recon18 grid=$g /"
done | tee -a $O/grid_sweep.log
