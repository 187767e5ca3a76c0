#!/bin/bash
# round 3, GPU call C: in-order queue version of the persistent kernel; reunitarisation tests; PMC
mkdir -p gpurun_out/r03_c gpurun_out/r03
O=gpurun_out/r03_c
timeout 900 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_reunit.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 6 $O/pytest.log
timeout 300 python scripts/r03/pipe_probe.py --mixed 1 --cg 100 > $O/pipe_probe.log 2>&1
cat $O/pipe_probe.log
for per in 1 2 3; do timeout 100 python scripts/dslash_probe.py --reps 100 --set dslash_pipe=1 --set pipe_per_cu=$per; done 2>&1 | tee $O/per_cu.log
bash scripts/r03/pmc_ab.sh pipeq12 dslash_pipe=1 > $O/pmc_pipeq12.log 2>&1
tail -n 25 $O/pmc_pipeq12.log
timeout 300 python scripts/r03/drift_probe.py --lattice 16,16,16,32 --steps 20 md_reunitarize=1 > $O/drift_16_reunit.log 2>&1; tail -n 4 $O/drift_16_reunit.log
