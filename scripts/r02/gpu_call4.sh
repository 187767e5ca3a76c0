#!/bin/bash
set -x
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call4.log
: > $L
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest4.log 2>&1; tail -5 gpurun_out/r02/pytest4.log >> $L
timeout 300 python scripts/r02/check_variants.py --lattice 16,16,16,32 --time 0 --variants 4,5 >> $L 2>&1
timeout 600 python scripts/r02/check_variants.py --lattice 32,32,32,64 --variants 5 --nts 0,5 >> $L 2>&1
for recon in 18 12; do for nt in "0 0" "1 1"; do set -- $nt
bash scripts/r02/pmc_traffic.sh pmc_r${recon}_nt$1$2 --set gauge_recon=$recon --set nt_gauge=$1 --set nt_store=$2 >> $L 2>&1
done; done
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/r02/bench4.json 2>gpurun_out/r02/bench4.err; cat gpurun_out/r02/bench4.json >> $L
grep -v "^+" $L | tail -60
