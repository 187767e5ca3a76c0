#!/bin/bash
mkdir -p gpurun_out/r02
L=gpurun_out/r02/call13.log; : > $L
timeout 900 python -m pytest tests/test_gpu_solver_edges.py tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_md.py tests/test_gpu_clover.py -x -q >> $L 2>&1
for d in 0 1; do for recon in 12 18; do
python scripts/dslash_probe.py --reps 100 --warm 10 --cg 200 --set gauge_recon=$recon --set cg_defer_x=$d >> $L 2>&1
done; done
tail -14 $L
