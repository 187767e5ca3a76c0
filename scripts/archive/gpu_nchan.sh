#!/bin/bash
cd "$(dirname "$0")/../.."
export LQCD_BENCH_FORCE_DIST=1 LQCD_FORCE_PARTITION=14
run() { timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --lattice 32,16,16,32 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['halo_phases_ms_max_over_ranks']; print('$1', round(d['value'],1), round(1e3*d['dslash_ms'],1), 'exch', round(1e3*p['exchange_after_pack'],1), 'int', round(1e3*p['interior'],1), 'tot', round(1e3*p['total_synchronised'],1))"; }
run default
for n in 4 8 16 32; do NCCL_MIN_P2P_NCHANNELS=$n NCCL_MAX_P2P_NCHANNELS=$n run "p2p_nchannels=$n"; done
NCCL_MIN_NCHANNELS=16 run "min_nchannels=16"
NCCL_PROTO=Simple run "proto=simple"
NCCL_PROTO=LL128 run "proto=ll128"
