#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 "$@" 2>&1 | grep -E "^dslash|^cg" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
for rep in 1 2; do
run v2 --set dslash_variant=2
run v3_p2 --set dslash_variant=3
run v3_p1 --set dslash_variant=3 --set persist_per_cu=1
run v3_p3 --set dslash_variant=3 --set persist_per_cu=3
run v3_p4 --set dslash_variant=3 --set persist_per_cu=4
done
run v3_cg --set dslash_variant=3 --cg 100
run v2_cg --set dslash_variant=2 --cg 100
run v3_small --lattice 16,16,16,32 --set dslash_variant=3
run v3_ns8 --set dslash_variant=3 --set xcd_nsub=8 --set xcd_ysplit=1
run v3_ns8ys2 --set dslash_variant=3 --set xcd_nsub=8 --set xcd_ysplit=2
run v3_ns32ys8 --set dslash_variant=3 --set xcd_nsub=32 --set xcd_ysplit=8
