#!/bin/bash
# round 3, GPU call B: persistent kernel -- parity again, A/B of the schedule variants (separate builds), PMC passes
mkdir -p gpurun_out/r03_b gpurun_out/r03
O=gpurun_out/r03_b
timeout 900 python -m pytest tests/test_gpu_pipe.py -x -q -m gpu > $O/pytest_pipe.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pipe.log
tail -4 $O/pytest_pipe.log
for lib in "" _dearly _nopf; do
  echo "== lib$lib" >> $O/pipe_probe.log
  LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip$lib.so timeout 300 python scripts/r03/pipe_probe.py --mixed 0 --cg 100 >> $O/pipe_probe.log 2>&1
done
cat $O/pipe_probe.log
for per in 1 2 3; do timeout 100 python scripts/dslash_probe.py --reps 100 --set dslash_pipe=1 --set pipe_per_cu=$per; done 2>&1 | tee $O/per_cu.log
bash scripts/r03/pmc_ab.sh plain12 > $O/pmc_plain12.log 2>&1
bash scripts/r03/pmc_ab.sh pipe12 dslash_pipe=1 > $O/pmc_pipe12.log 2>&1
tail -30 $O/pmc_plain12.log $O/pmc_pipe12.log
