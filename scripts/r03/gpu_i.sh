#!/bin/bash
# round 3, GPU call I: whole GPU suite (no -x), multi-chunk pipelined workgroups A/B
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r03_i; rm -rf $O; mkdir -p $O gpurun_out/r03
timeout 400 python scripts/r03/pipe_probe.py --mixed 0 --cg 200 > $O/pipe_probe.log 2>&1; cat $O/pipe_probe.log
bash scripts/r03/pmc_ab.sh pair12 dslash_pipe=3 > $O/pmc.log 2>&1
grep "FETCH\|TCC_" gpurun_out/r03/pmc_pair12.csv
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 12 $O/pytest.log
