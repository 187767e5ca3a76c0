#!/bin/bash
# round 3, GPU call E: fused tails of the partitioned CG -- parity, proxy scaling A/B, timelines
mkdir -p gpurun_out/r03_e
O=gpurun_out/r03_e
timeout 1200 python -m pytest tests/test_gpu_halo_fuse.py tests/test_gpu_pipe.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 15 $O/pytest.log
for fuse in 0 1 2 3; do
  for mode in 1 0 2; do
    LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 100 --warm 20 --cg 400 --set halo_fuse=$fuse --set halo_stream_mode=$mode 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 fuse=$fuse mode=$mode /"; echo
  done
done 2>&1 | tee $O/proxy_n8.log
timeout 200 python scripts/dslash_probe.py --lattice 32,32,32,64 --reps 100 --warm 20 --cg 300 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=1 /" | tee -a $O/proxy_n8.log; echo
