#!/bin/bash
cd "$(dirname "$0")/../.."
export LQCD_BENCH_FORCE_DIST=1
echo "--- torchrun, 1 rank, distributed branch, unpartitioned"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-700
echo "--- torchrun, 1 rank, distributed branch, self-partition y,z,t on the 8-GPU local volume"
LQCD_FORCE_PARTITION=14 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --lattice 32,16,16,32 2>&1 | tail -2 | cut -c1-700
