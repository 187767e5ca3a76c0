#!/bin/bash
# fp32 paired-component layout: correctness (mixed tests) then timing + kernel trace of one mixed solve
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_fuzz.py -x -q > gpurun_out/r02/pytest25.log 2>&1; tail -5 gpurun_out/r02/pytest25.log
python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-16 2>&1 | tail -2
python scripts/mixed_probe.py 32,32,32,64 Staggered 1e-16 2>&1 | tail -2
python scripts/mixed_probe.py 32,32,32,64 WilsonClover 1e-16 2>&1 | tail -2
bash scripts/gpu_mixed_trace.sh 2>&1 | tail -30
