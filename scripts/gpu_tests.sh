#!/bin/bash
set -x
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -25 gpurun_out/pytest_gpu.log
timeout 900 python scripts/bench_extra.py C2 C4 IVF_S --beams 8 --steps 2 2>/dev/null | grep encode
QINCO_TABLE_VALU=1 timeout 900 python scripts/bench_extra.py C2 C4 IVF_S --beams 8 --steps 2 2>/dev/null | grep encode
