#!/bin/bash
# round 3, GPU call G: scalar-addressing kernel (dslash_pipe = 2) parity + A/B; fused tails third version; PMC of mode 2
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r03_g; rm -rf $O; mkdir -p $O gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_halo_fuse.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 8 $O/pytest.log
timeout 400 python scripts/r03/pipe_probe.py --mixed 1 --cg 200 > $O/pipe_probe.log 2>&1; cat $O/pipe_probe.log
for fuse in 0 1 2 3; do
  LQCD_FORCE_PARTITION=14 timeout 200 python scripts/dslash_probe.py --lattice 32,16,16,32 --selfcomm 1 --reps 100 --warm 20 --cg 400 --set halo_fuse=$fuse --set halo_stream_mode=1 2>&1 | grep -E "^dslash|^cg" | tr '\n' ' ' | sed "s/^/N=8 fuse=$fuse mode=1 /"; echo
done 2>&1 | tee $O/proxy_n8.log
bash scripts/r03/pmc_ab.sh sdir12 dslash_pipe=2 > $O/pmc.log 2>&1
grep "FETCH\|TCC\|WAIT\|WAVE_CYC\|INSTS" gpurun_out/r03/pmc_sdir12.csv
