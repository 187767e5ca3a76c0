#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/mixed; mkdir -p $O; export TMPDIR=/tmp
timeout 200 python scripts/mixed_probe.py 2>&1 | tail -2
timeout 200 python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-19 2>&1 | tail -2
timeout 200 python scripts/mixed_probe.py 48,48,48,96 Staggered 1e-12 2>&1 | tail -2
timeout 200 python scripts/mixed_probe.py 16,16,16,32 Wilson 1e-19 2>&1 | tail -2
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o m -- python $R/scripts/mixed_probe.py > $O/trace.log 2>&1)
f=$(find $O/trace -name '*kernel_stats.csv' | head -1); head -14 $f | cut -c1-200
find $O -name '*kernel_trace.csv' -size +20M -delete
