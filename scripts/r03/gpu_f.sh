#!/bin/bash
# round 3, GPU call F: fused tails, second version (write-through partials, pack blocks first) -- parity, proxy A/B, timelines
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r03_f; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_halo_fuse.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 5 $O/pytest.log
for fuse in 0 1 2 3; do
  LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 100 --warm 20 --cg 400 --set halo_fuse=$fuse --set halo_stream_mode=1 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 fuse=$fuse mode=1 /"; echo
done 2>&1 | tee $O/proxy_n8.log
for fuse in 0 3; do
  (cd /tmp && LQCD_FORCE_PARTITION=14 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/f$fuse -o t -- python $R/scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 20 --warm 5 --cg 60 --set halo_stream_mode=1 --set halo_fuse=$fuse > $O/f$fuse.log 2>&1)
  f=$(find $O/f$fuse -name "*kernel_trace.csv" | head -1)
  echo "== halo_fuse $fuse (halo_stream_mode 1)"; grep -E "^cg" $O/f$fuse.log; python scripts/timeline.py $f cg_update_odd -3 2>&1 | head -40
done 2>&1 | tee $O/timeline.log
