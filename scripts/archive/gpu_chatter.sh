#!/bin/bash
cd "$(dirname "$0")/../.."
export LQCD_FORCE_PARTITION=14
MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 LQCD_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 5 > gpurun_out/chat_stdout.txt 2> gpurun_out/chat_stderr.txt
echo "rc=$?"
echo "--- stdout (last 3 lines, cut)"; tail -3 gpurun_out/chat_stdout.txt | cut -c1-300
echo "--- stderr (last 12 lines)"; tail -12 gpurun_out/chat_stderr.txt | cut -c1-200
