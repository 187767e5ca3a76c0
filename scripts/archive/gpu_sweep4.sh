#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); mkdir -p gpurun_out/sweep4; export TMPDIR=/tmp
run() {
  label=$1; shift
  line=$(python scripts/dslash_probe.py --reps 100 --warm 10 "$@" 2>&1 | grep "^dslash" | sed 's/.*ms=/ms=/')
  if [ -n "$PMC" ]; then (cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/sweep4/$label -o p -- python $R/scripts/dslash_probe.py --reps 3 --warm 1 "$@" > /dev/null 2>&1); fi
  echo "$label | $line"
}
PMC=
for rm in 0 1 2; do
  run v2_r${rm} --set dslash_variant=2 --set xcd_remap=$rm
  run v1_r${rm} --set dslash_variant=1 --set xcd_remap=$rm
done
run v2_r2_pad16 --set dslash_variant=2 --set xcd_remap=2 --set lds_pad_kb=16
run v2_r2_pad32 --set dslash_variant=2 --set xcd_remap=2 --set lds_pad_kb=32
run v0_b64_r2_pad20 --set dslash_block=64 --set xcd_remap=2 --set lds_pad_kb=20
run v0_b64_r2_pad0 --set dslash_block=64 --set xcd_remap=2
run v0_b128_r2_pad0 --set dslash_block=128 --set xcd_remap=2
run v0_b128_r2_pad40 --set dslash_block=128 --set xcd_remap=2 --set lds_pad_kb=40
run v2_r2_dag --set dslash_variant=2 --set xcd_remap=2 --dagger 1
run small16_v2_r2 --lattice 16,16,16,32 --set dslash_variant=2 --set xcd_remap=2
run big48_v2_r2 --lattice 48,48,48,48 --set dslash_variant=2 --set xcd_remap=2
python scripts/dslash_probe.py --reps 20 --cg 50 --set dslash_variant=2 --set xcd_remap=2 | tail -1
python scripts/dslash_probe.py --reps 20 --cg 50 --set dslash_variant=1 --set xcd_remap=2 | tail -1
