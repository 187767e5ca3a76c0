#!/bin/bash
# usage: gpu_profile_round.sh <tag>     (on the GPU box, one gpurun call)
# rocprofv3 evidence of one round -> gpurun_out/profile_<tag>/ ; scripts/collect_profiles.py condenses it into profiles/<tag>_* and
# regenerates the "measured" table of profiles/README.md FROM those files.
#   1. bench.py un-profiled (its own HIP-event figures, live PMC traffic)
#   2. the same command under rocprofv3 --kernel-trace --stats (kernel averages must agree with 1.)
#   3. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | TCC hit/miss/EA) over the Dslash probe, default links and all 18 reals
cd "$(dirname "$0")/.."
TAG=${1:-r02}
R=$(pwd); O=$R/gpurun_out/profile_$TAG; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --steps 400 --warmup 20 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-pmc > $O/trace_bench.json 2> $O/trace.err)
for recon in 12 18; do
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $O/pmc_r${recon}_$n -o p -- python $R/scripts/dslash_probe.py --reps 5 --warm 1 --set gauge_recon=$recon > $O/pmc_r${recon}_$n.log 2>&1) || echo "pmc $recon $n failed"
done; done
ls $O
