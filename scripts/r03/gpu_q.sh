#!/bin/bash
# call Q: one-launch CG for launch-bound lattices (cg_persist.hip): tests, then the 8^4 staggered solve with and without it
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_q; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_cg_persist.py -x -q > $O/test.log 2>&1; tail -15 $O/test.log
timeout 120 python scripts/r03/small_cg_probe.py > $O/probe.log 2>&1; cat $O/probe.log
timeout 600 python -m pytest tests/test_gpu_solver_edges.py tests/test_gpu_md_staggered.py tests/test_gpu_rhmc.py -x -q > $O/others.log 2>&1; tail -3 $O/others.log
