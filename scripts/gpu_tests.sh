#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log
