#!/bin/bash
# round 3, GPU call H: the whole GPU suite on the new defaults (dslash_pipe = 2, halo_fuse = 2, md_reunitarize = 1), then bench.py
cd "$(dirname "$0")/../.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/r03_h; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -n 30 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 2500 $O/bench.json; tail -n 5 $O/bench.err
