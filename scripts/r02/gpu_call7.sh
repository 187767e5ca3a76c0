#!/bin/bash
mkdir -p gpurun_out/r02
timeout 1700 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02/pytest7.log 2>&1; tail -25 gpurun_out/r02/pytest7.log
timeout 900 python bench.py > gpurun_out/r02/bench7.json 2> gpurun_out/r02/bench7.err; tail -c 3000 gpurun_out/r02/bench7.json; tail -5 gpurun_out/r02/bench7.err
