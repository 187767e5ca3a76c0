#!/bin/bash
# the driver's N > 1 launch line with one rank: torch.distributed.run + the distributed control path + self-partitioned halos
cd "$(dirname "$0")/.."
export LQCD_BENCH_FORCE_DIST=1 LQCD_FORCE_PARTITION=14
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 5 > gpurun_out/torchrun1.out 2> gpurun_out/torchrun1.err
echo "rc=$?"; echo "stdout lines: $(wc -l < gpurun_out/torchrun1.out)"; tail -1 gpurun_out/torchrun1.out | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'n_gpus', 'ms_per_step', 'halo_stream_mode_rank0', 'halo_phases_ms_max_over_ranks', 'allreduce_latency_us')})
print(d['config'])"
tail -3 gpurun_out/torchrun1.err | cut -c1-200
