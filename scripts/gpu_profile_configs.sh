#!/bin/bash
# rocprofv3 kernel stats of scripts/bench_configs.py (staggered, even-odd BiCGStab, multi-shift, force sweep, MD-step kernels) and of the clover probes
cd "$(dirname "$0")/.."
R=$(pwd); export TMPDIR=/tmp; O=$R/gpurun_out/profile_configs; mkdir -p $O
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg -o cfg -- python $R/scripts/bench_configs.py > $O/cfg.out 2>&1)
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/clov -o clov -- python $R/scripts/eo_probe.py 32,32,32,64 > $O/clov.out 2>&1)
ls $O/cfg $O/clov | head; head -25 $O/cfg/cfg_kernel_stats.csv | cut -c1-200
