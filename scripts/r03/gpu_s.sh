#!/bin/bash
# call S: halo schedule 3 (one stream, no overlap, no join) against schedules 1 / 2 on the N = 8, 4, 2 local volumes (RCCL to self), and its tests
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_halo_fuse.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for vol in "32,16,16,32 14 8" "32,32,16,32 12 4" "32,32,32,32 8 2"; do set -- $vol
for rep in 1 2; do for mode in 1 2 3 -1; do
  LQCD_FORCE_PARTITION=$2 timeout 200 python scripts/dslash_probe.py --lattice $1 --selfcomm 1 --reps 100 --warm 20 --cg 400 --set halo_fuse=2 --set halo_stream_mode=$mode 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=$3 mode=$mode /"; echo
done; done; done 2>&1 | tee $O/proxy.log
