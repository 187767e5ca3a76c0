#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_recon12.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/compress -o c -- python $R/scripts/dslash_probe.py --reps 5 --warm 1 > /dev/null 2>&1)
grep -h "gauge_compress12\|Name" $R/gpurun_out/compress/*kernel_stats.csv $R/gpurun_out/compress/*/*kernel_stats.csv 2>/dev/null | cut -c1-200
