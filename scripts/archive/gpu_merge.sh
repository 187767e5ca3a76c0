#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 600 python -m pytest tests -m gpu -q -x -k "rccl or self_partition or partitioned or mixed" 2>&1 | tail -2
export LQCD_FORCE_PARTITION=14
for m in 1 0; do
LQCD_SET="halo_merge=$m" timeout 200 python scripts/mixed_probe.py 32,16,16,32 Wilson 1e-16 2>&1 | tail -2 | sed "s/^/selfcomm? merge=$m /"
done
