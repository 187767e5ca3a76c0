#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests -m gpu -q -x -k "rccl or self_partition or partitioned or mixed or graph" 2>&1 | tail -2
export LQCD_FORCE_PARTITION=14
for m in -1 0 1; do
timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 200 --warm 20 --cg 400 --set halo_stream_mode=$m 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/mode=$m /"; echo
done
LQCD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','halo_stream_mode_rank0','halo_phases_ms_max_over_ranks')})"
