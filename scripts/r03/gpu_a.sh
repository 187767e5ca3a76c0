#!/bin/bash
# round 3, GPU call A: parity of the persistent kernel, A/B timing, unitarity drift
mkdir -p gpurun_out/r03_a
O=gpurun_out/r03_a
timeout 900 python -m pytest tests/test_gpu_pipe.py -x -q -m gpu > $O/pytest_pipe.log 2>&1; echo "pytest rc=$?" >> $O/pytest_pipe.log
tail -5 $O/pytest_pipe.log
timeout 600 python scripts/r03/pipe_probe.py > $O/pipe_probe.log 2>&1; tail -12 $O/pipe_probe.log
timeout 300 python scripts/r03/drift_probe.py --lattice 16,16,16,32 --steps 20 > $O/drift_16.log 2>&1; tail -8 $O/drift_16.log
timeout 300 python scripts/r03/drift_probe.py --lattice 32,32,32,64 --steps 6 > $O/drift_32.log 2>&1; tail -8 $O/drift_32.log
