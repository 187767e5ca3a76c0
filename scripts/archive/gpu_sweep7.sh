#!/bin/bash
cd "$(dirname "$0")/../.."
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 "$@" 2>&1 | grep "^dslash" | sed 's/.*ms=/ms=/'); echo "$label | $line"; }
V2="--set dslash_variant=2 --set xcd_remap=2 --set xcd_nsub=16 --set xcd_ysplit=4"
for rep in 1 2; do
run base $V2
run dbg1_no_xspinor $V2 --set dbg=1
run dbg2_no_xlink $V2 --set dbg=2
run dbg3_no_matvec $V2 --set dbg=3
done
run base_pad16 $V2 --set lds_pad_kb=16
