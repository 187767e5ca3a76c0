#!/bin/bash
# refresh of the secondary measurements with the round's final code
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r02
python scripts/bench_configs.py > gpurun_out/r02/bench_configs.log 2>&1; tail -c 600 gpurun_out/r02/bench_configs.log
for k in Wilson Staggered WilsonClover; do python scripts/mixed_probe.py 32,32,32,64 $k 1e-16 2>&1 | tail -2; done > gpurun_out/r02/mixed_precision.log
python scripts/mixed_probe.py 48,48,48,96 Staggered 1e-12 2>&1 | tail -2 >> gpurun_out/r02/mixed_precision.log
cat gpurun_out/r02/mixed_precision.log
python scripts/r02/md_probe.py > gpurun_out/r02/md_probe.log 2>&1; tail -5 gpurun_out/r02/md_probe.log | cut -c1-300
