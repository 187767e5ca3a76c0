#!/bin/bash
# kernel timeline of a partitioned CG iteration (RCCL send/recv to self) at the 8-GPU local volume, and of the 1-GPU CG
cd "$(dirname "$0")/../.."
R=$(pwd); O=$R/gpurun_out/timeline; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && LQCD_FORCE_PARTITION=14 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/n8 -o t -- python $R/scripts/dslash_probe.py --lattice 32,16,16,32 --reps 20 --warm 3 --cg 60 --selfcomm 1 > $O/n8.log 2>&1)
tail -2 $O/n8.log
f=$(find $O/n8 -name '*kernel_trace.csv' | head -1); python scripts/timeline.py $f cg_update_xp -3 | tee $O/n8_timeline.txt
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/n1 -o t -- python $R/scripts/dslash_probe.py --reps 20 --warm 3 --cg 60 > $O/n1.log 2>&1)
tail -2 $O/n1.log
f=$(find $O/n1 -name '*kernel_trace.csv' | head -1); python scripts/timeline.py $f cg_update_xp -3 | tee $O/n1_timeline.txt
find $O -name '*.csv' -size +20M -delete
