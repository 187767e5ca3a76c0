#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for f in 2 1 0; do python scripts/dslash_probe.py --reps 100 --warm 10 --cg 100 --set cg_fused=$f | tr '\n' ' '; echo " [cg_fused=$f]"; done
LQCD_PAD_CHUNKS=3 python scripts/dslash_probe.py --reps 100 --warm 10 --cg 100 | tr '\n' ' '; echo " [pad3]"
python scripts/dslash_probe.py --reps 100 --warm 10 --cg 100 --set dslash_variant=1 | tr '\n' ' '; echo " [v1]"
timeout 600 python bench.py --steps 100 --warmup 10 | tail -1 > gpurun_out/bench2.json; cat gpurun_out/bench2.json
