#!/bin/bash
# round 2, call 1: clean (un-profiled) timings of the traffic ablations of the dirsplit kernel, 18- and 12-real links
set -x
mkdir -p gpurun_out/r02
L=gpurun_out/r02/diag1.log
: > $L
for recon in 18 12; do
for dbg in 0 983040 61440 3840 986880 1044480 1048320; do
  python scripts/dslash_probe.py --reps 200 --warm 20 --set gauge_recon=$recon --set dbg=$dbg >> $L 2>&1
done
done
# occupancy limiter on the 12-real kernel
for pad in 0 8 16 32; do
  python scripts/dslash_probe.py --reps 200 --warm 20 --set gauge_recon=12 --set lds_pad_kb=$pad >> $L 2>&1
done
python scripts/stream_probe.py >> $L 2>&1
cat $L
