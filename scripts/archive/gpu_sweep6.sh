#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); mkdir -p gpurun_out/sweep6; export TMPDIR=/tmp
run() {
  label=$1; shift
  line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 "$@" 2>&1 | grep "^dslash" | sed 's/.*ms=/ms=/')
  pm=""
  if [ -n "$PMC" ]; then
    (cd /tmp && timeout 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/sweep6/$label -o p -- python $R/scripts/dslash_probe.py --reps 3 --warm 1 "$@" > /dev/null 2>&1)
    pm=$(python scripts/tcc_summary.py gpurun_out/sweep6/$label)
  fi
  echo "$label | $line | $pm"
}
PMC=1
V2="--set dslash_variant=2 --set xcd_remap=2"
run ns8_ys1 $V2 --set xcd_nsub=8
run ns8_ys2 $V2 --set xcd_nsub=8 --set xcd_ysplit=2
run ns16_ys1 $V2 --set xcd_nsub=16
run ns16_ys2 $V2 --set xcd_nsub=16 --set xcd_ysplit=2
run ns16_ys4 $V2 --set xcd_nsub=16 --set xcd_ysplit=4
run ns32_ys2 $V2 --set xcd_nsub=32 --set xcd_ysplit=2
run ns32_ys4 $V2 --set xcd_nsub=32 --set xcd_ysplit=4
run ns32_ys8 $V2 --set xcd_nsub=32 --set xcd_ysplit=8
run ns64_ys4 $V2 --set xcd_nsub=64 --set xcd_ysplit=4
run ns64_ys8 $V2 --set xcd_nsub=64 --set xcd_ysplit=8
PMC=
run ns16_ys4_dag $V2 --set xcd_nsub=16 --set xcd_ysplit=4 --dagger 1
run v1_ns16_ys4 --set dslash_variant=1 --set xcd_remap=2 --set xcd_nsub=16 --set xcd_ysplit=4
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "variant or full_size" 2>&1 | tail -3
