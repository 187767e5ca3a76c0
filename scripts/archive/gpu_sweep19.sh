#!/bin/bash
cd "$(dirname "$0")/../.."
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 300 --warm 30 "$@" 2>&1 | grep -E "^dslash" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
for v in 1 2; do
 for cfg in "8 1" "8 2" "16 1" "16 2" "16 4" "32 2" "32 4" "32 8" "64 8"; do
  set -- $cfg
  run v${v}_ns$1_ys$2 --set dslash_variant=$v --set xcd_nsub=$1 --set xcd_ysplit=$2
 done
done
run v1_remap1 --set dslash_variant=1 --set xcd_remap=1
run v1_remap0 --set dslash_variant=1 --set xcd_remap=0
for pad in 0 3 7; do LQCD_PAD_CHUNKS=$pad run v1_padchunks$pad --set dslash_variant=1; done
run v1_ldspad8 --set dslash_variant=1 --set lds_pad_kb=8
run v1_dag --set dslash_variant=1 --dagger 1
