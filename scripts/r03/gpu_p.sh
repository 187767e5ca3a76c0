#!/bin/bash
# call P: the whole GPU suite on the round's tree, the opt-in variants build through its tests, the round's profile evidence, the other BASELINE configs
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_p; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/suite.log 2>&1; tail -4 $O/suite.log
LQCD_HIP_LIB=$(pwd)/latticeqcd.jl_amd/csrc/liblqcd_hip_variants.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "variant" > $O/variants.log 2>&1; tail -3 $O/variants.log
bash scripts/gpu_profile_round.sh r03 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
timeout 900 python scripts/bench_configs.py > $O/bench_configs.log 2>&1; tail -30 $O/bench_configs.log
