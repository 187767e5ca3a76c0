#!/bin/bash
# call T: kernel timeline of the partitioned CG at the N = 8 local volume under halo schedule 3 (one stream) and 1, halo_fuse = 2, RCCL to self
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r03_t; rm -rf $O; mkdir -p $O
for mode in 3 1; do
  (cd /tmp && LQCD_FORCE_PARTITION=14 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$mode -o t -- python $R/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 --cg 60 --set halo_fuse=2 --set halo_stream_mode=$mode > $O/m$mode.log 2>&1)
  f=$(find $O/tr$mode -name "*kernel_trace.csv" | head -1)
  echo "== halo_stream_mode $mode"; grep -E "^cg" $O/m$mode.log; python scripts/timeline.py $f cg_update_odd -3 2>&1 | head -40
done 2>&1 | tee $O/timeline.log
