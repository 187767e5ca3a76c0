#!/bin/bash
cd "$(dirname "$0")/.."
export LQCD_BENCH_FORCE_DIST=1 LQCD_FORCE_PARTITION=14
for m in 1 0 1 0; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$m bench.py --gpus 1 --lattice 32,16,16,32 --no-cpu-baseline --set halo_merge=$m 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('merge=$m', round(d['value'],1), d['dslash_ms'], d['halo_phases_ms_max_over_ranks'])"
done
unset LQCD_BENCH_FORCE_DIST
timeout 300 python -m pytest tests -m gpu -q -x -k "rccl or self_partition or partitioned" 2>&1 | tail -2
