#!/bin/bash
# rocprofv3 kernel-trace/stats of bench.py + HBM-traffic PMC passes of the Dslash kernel -> gpurun_out/profile_r01/
cd "$(dirname "$0")/.."
R=$(pwd); O=$R/gpurun_out/profile_r01f; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --steps 100 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/trace_bench.json 2> $O/trace.err)
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 120 rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$n -o p -- python $R/scripts/dslash_probe.py --reps 5 --warm 1 > $O/pmc_$n.log 2>&1) || echo "pmc $n failed"
done
python scripts/summarize_prof.py $O 2>&1 | head -60
ls $O/trace
