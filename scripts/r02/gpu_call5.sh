#!/bin/bash
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest5.log 2>&1; tail -30 gpurun_out/r02/pytest5.log
