#!/bin/bash
# call O: full-size short trajectories (configs 4/5 on one GPU), fp32 one-site-per-lane path back on variant 1, bench line with the all-cores oracle figure
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s > $O/fullsize.log 2>&1; tail -5 $O/fullsize.log
timeout 600 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_pair32.py tests/test_gpu_pipe.py tests/test_gpu_md_mixed.py -x -q > $O/mixed.log 2>&1; tail -3 $O/mixed.log
timeout 300 python scripts/r03/mixed_ab.py mixed_pair32=0 > $O/mixed_ab.log 2>&1; tail -3 $O/mixed_ab.log
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
