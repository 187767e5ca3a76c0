#!/bin/bash
cd "$(dirname "$0")/../.."
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 200 --warm 20 "$@" 2>&1 | grep -E "^dslash|^cg" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
for rep in 1 2 3; do
run v2_buf --set dslash_variant=2
run v3_buf --set dslash_variant=3
run v1 --set dslash_variant=1
done
timeout 600 python -m pytest tests -m gpu -q -x -k "dirsplit or fixture or full_size" 2>&1 | tail -2
