#!/bin/bash
# the three halo schedules (and the auto choice) at the local volumes of N = 8, 4, 2 GPUs; self-partition proxy on one GPU
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests -m gpu -q -x -k "rccl or self_partition or partitioned" 2>&1 | tail -2
run() { LQCD_FORCE_PARTITION=$1 timeout 200 python scripts/dslash_probe.py --lattice $2 --selfcomm 1 --reps 200 --warm 20 --cg 400 --set halo_stream_mode=$3 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/mask=$1 mode=$3 /" | cut -c1-200; echo; }
for m in 0 1 2 -1; do run 14 32,16,16,32 $m; done
for m in 0 1 2 -1; do run 12 32,32,16,32 $m; done
for m in 0 1 2 -1; do run 8 32,32,32,32 $m; done
