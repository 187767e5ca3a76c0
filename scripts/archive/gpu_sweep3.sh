#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); mkdir -p gpurun_out/sweep3; export TMPDIR=/tmp
run() {  # label, probe args...
  label=$1; shift
  line=$(python scripts/dslash_probe.py --reps 100 --warm 10 "$@" 2>&1 | grep "^dslash" | sed 's/.*ms=/ms=/')
  if [ -n "$PMC" ]; then (cd /tmp && rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/sweep3/$label -o p -- python $R/scripts/dslash_probe.py --reps 3 --warm 1 "$@" > /dev/null 2>&1); fi
  echo "$label | $line"
}
PMC=1
for rm in 1 2; do
  run v0_b64_r${rm}_pad20 --set dslash_block=64 --set xcd_remap=$rm --set lds_pad_kb=20
  run v0_b64_r${rm}_pad40 --set dslash_block=64 --set xcd_remap=$rm --set lds_pad_kb=40
  run v0_b128_r${rm}_pad40 --set dslash_block=128 --set xcd_remap=$rm --set lds_pad_kb=40
  run v0_b128_r${rm}_pad80 --set dslash_block=128 --set xcd_remap=$rm --set lds_pad_kb=80
  run v0_b256_r${rm}_pad80 --set dslash_block=256 --set xcd_remap=$rm --set lds_pad_kb=80
done
for rm in 0 1 2; do for pad in 0 32 64; do
  run v1_r${rm}_pad${pad} --set dslash_variant=1 --set xcd_remap=$rm --set lds_pad_kb=$pad
done; done
PMC=
run small16_v1_r2 --lattice 16,16,16,32 --set dslash_variant=1 --set xcd_remap=2
run small16_v0 --lattice 16,16,16,32 --set dslash_block=64 --set lds_pad_kb=20
run mid24_v1_r2 --lattice 24,24,24,48 --set dslash_variant=1 --set xcd_remap=2
run big48_v1_r2 --lattice 48,48,48,48 --set dslash_variant=1 --set xcd_remap=2
