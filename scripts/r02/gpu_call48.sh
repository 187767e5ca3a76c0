#!/bin/bash
# per-iteration timeline of the CG at the N = 8 local volume (self-partition y,z,t; RCCL to self), current code, each halo schedule
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r02/tl; rm -rf $O; mkdir -p $O
for mode in 1 2 0; do
  (cd /tmp && LQCD_FORCE_PARTITION=14 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/m$mode -o t -- python $R/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 --cg 60 --set halo_stream_mode=$mode > $O/m$mode.log 2>&1)
  f=$(find $O/m$mode -name "*kernel_trace.csv" | head -1)
  echo "== halo_stream_mode $mode"; grep -E "^cg" $O/m$mode.log; python scripts/timeline.py $f cg_update_odd -3 2>&1 | head -40
done
