#!/bin/bash
cd "$(dirname "$0")/../.."
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 "$@" 2>&1 | grep "^dslash" | sed 's/.*ms=/ms=/'); echo "$label | $line"; }
V2="--set dslash_variant=2 --set xcd_remap=2 --set xcd_nsub=16 --set xcd_ysplit=4"
for pad in 0 1 3 17; do
  export LQCD_PAD_CHUNKS=$pad
  run pad${pad}_v2 $V2
  run pad${pad}_v2_ns8 --set dslash_variant=2 --set xcd_remap=2
  run pad${pad}_v1 --set dslash_variant=1 --set xcd_remap=2 --set xcd_nsub=16 --set xcd_ysplit=4
  run pad${pad}_v0_b64_pad20 --set dslash_block=64 --set xcd_remap=2 --set lds_pad_kb=20
  run pad${pad}_v0_b128 --set dslash_block=128 --set xcd_remap=1
done
export LQCD_PAD_CHUNKS=1
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
export LQCD_PAD_CHUNKS=0
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
