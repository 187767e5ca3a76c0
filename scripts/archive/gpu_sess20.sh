#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
run() { label=$1; shift; line=$(timeout 120 python scripts/dslash_probe.py --reps 300 --warm 30 --cg 100 "$@" 2>&1 | grep -E "^dslash|^cg" | sed 's/.*ms=/ms=/' | tr '\n' ' '); echo "$label | $line"; }
for rep in 1 2 3; do
run v1
run v2 --set dslash_variant=2
done
run stag8 --lattice 8,8,8,8 --kind Staggered
run stag32 --lattice 32,32,32,32 --kind Staggered
