#!/bin/bash
cd "$(dirname "$0")/../.."
timeout 400 python -m pytest tests/test_gpu_mixed.py tests/test_gpu_recon12.py -m gpu -q -x 2>&1 | tail -3
timeout 200 python scripts/mixed_probe.py 2>&1 | tail -2
timeout 200 python scripts/mixed_probe.py 32,32,32,64 Wilson 1e-19 2>&1 | tail -2
timeout 200 python scripts/mixed_probe.py 48,48,48,96 Staggered 1e-12 2>&1 | tail -2
