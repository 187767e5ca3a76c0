#!/bin/bash
# Sweep of kernel variants with L2 hit-rate counters.
cd "$(dirname "$0")/../.."
R=$(pwd); mkdir -p gpurun_out/sweep2; export TMPDIR=/tmp
run() {  # label, probe args...
  label=$1; shift
  line=$(python scripts/dslash_probe.py --reps 100 --warm 10 "$@" 2>&1 | grep "^dslash" | sed 's/.*ms=/ms=/')
  (cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/sweep2/$label -o p -- python $R/scripts/dslash_probe.py --reps 3 --warm 1 "$@" > /dev/null 2>&1)
  pm=$(python scripts/summarize_prof.py gpurun_out/sweep2 2>/dev/null | grep -A4 "$label/" | grep -E "wilson" | awk '{printf "%s=%s ", $(NF-2), $(NF-1)}')
  echo "$label | $line | $pm"
}
run v0_b128 --set dslash_block=128
run v0_b256 --set dslash_block=256
run v0_b256_pad40 --set dslash_block=256 --set lds_pad_kb=40
run v0_b256_pad80 --set dslash_block=256 --set lds_pad_kb=80
run v0_b128_pad20 --set dslash_block=128 --set lds_pad_kb=20
run v0_b128_pad40 --set dslash_block=128 --set lds_pad_kb=40
run v0_b64_pad20 --set dslash_block=64 --set lds_pad_kb=20
run v1_remap1 --set dslash_variant=1
run v1_remap0 --set dslash_variant=1 --set xcd_remap=0
run v1_dag --set dslash_variant=1 --dagger 1
python scripts/dslash_probe.py --reps 50 --cg 50 --set dslash_variant=1 | tail -1
python scripts/dslash_probe.py --reps 50 --cg 50 --set dslash_block=256 | tail -1
