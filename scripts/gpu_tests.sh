#!/bin/bash
# GPU parity suite under the production kernels and under every A/B variant that is still compiled in
set -x
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
for v in "48,196" "48,92" "48,76" "36,12" "8,0"; do
  QINCO_MLP_VARIANT=$v timeout 1800 python -m pytest tests -m gpu -x -q -k "golden or oracle" > gpurun_out/pytest_gpu_$v.log 2>&1; tail -3 gpurun_out/pytest_gpu_$v.log
done
