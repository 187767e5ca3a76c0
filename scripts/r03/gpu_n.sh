#!/bin/bash
# round 3, GPU call N: staple sweep with two-row link loads -- tests, timing at 32^3x64; MD-related suites
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_n; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_reunit.py tests/test_gpu_md.py tests/test_gpu_md_partitioned.py tests/test_gpu_reference_callers.py tests/test_gpu_hmc_partitioned.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
for rep in 1 2; do
  timeout 200 python scripts/r02/md_probe.py staple_recon=1
  timeout 200 python scripts/r02/md_probe.py staple_recon=0
done 2>&1 | tee $O/md_probe.log
