#!/bin/bash
# what the driver does at round end: build check, smoke, GPU tests, default bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/final
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; tail -c 2500 gpurun_out/final/bench_n1.json; tail -3 gpurun_out/final/bench_n1.err
