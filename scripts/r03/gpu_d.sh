#!/bin/bash
# round 3, GPU call D: persistent kernel with within-chunk queue lookahead
mkdir -p gpurun_out/r03_d gpurun_out/r03
O=gpurun_out/r03_d
timeout 900 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_reunit.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
for lib in "" _dearly; do
  echo "== lib$lib" >> $O/pipe_probe.log
  LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip$lib.so timeout 300 python scripts/r03/pipe_probe.py --mixed 0 --cg 100 >> $O/pipe_probe.log 2>&1
done
cat $O/pipe_probe.log
bash scripts/r03/pmc_ab.sh pipeq12b dslash_pipe=1 > $O/pmc.log 2>&1
grep "FETCH\|TCC\|WAIT\|WAVE_CYC" gpurun_out/r03/pmc_pipeq12b.csv
