#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd); mkdir -p gpurun_out/sweep5; export TMPDIR=/tmp
run() {
  label=$1; shift
  line=$(timeout 120 python scripts/dslash_probe.py --reps 100 --warm 10 "$@" 2>&1 | grep "^dslash" | sed 's/.*ms=/ms=/')
  pm=""
  if [ -n "$PMC" ]; then
    (cd /tmp && timeout 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/sweep5/$label -o p -- python $R/scripts/dslash_probe.py --reps 3 --warm 1 "$@" > /dev/null 2>&1)
    pm=$(python scripts/tcc_summary.py gpurun_out/sweep5/$label)
  fi
  echo "$label | $line | $pm"
}
PMC=1
for ns in 8 16 32; do run v2_ns${ns} --set dslash_variant=2 --set xcd_remap=2 --set xcd_nsub=$ns; done
for ns in 8 16; do
  run v2_ns${ns}_nts --set dslash_variant=2 --set xcd_remap=2 --set xcd_nsub=$ns --set nt_store=1
  run v2_ns${ns}_ntg --set dslash_variant=2 --set xcd_remap=2 --set xcd_nsub=$ns --set nt_gauge=1
  run v2_ns${ns}_ntgs --set dslash_variant=2 --set xcd_remap=2 --set xcd_nsub=$ns --set nt_gauge=1 --set nt_store=1
done
PMC=
run small16_ntgs --lattice 16,16,16,32 --set dslash_variant=2 --set xcd_remap=2 --set nt_gauge=1 --set nt_store=1
run v1_ns16 --set dslash_variant=1 --set xcd_remap=2 --set xcd_nsub=16
run v0_b64_ns16_pad20 --set dslash_block=64 --set xcd_remap=2 --set xcd_nsub=16 --set lds_pad_kb=20
