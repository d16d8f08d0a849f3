#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or fresh or instance or variants or small_models or full_size" 2>&1 | tail -2
for v in "" "48,124"; do
  QINCO_MLP_VARIANT=$v timeout 600 python scripts/bench_extra.py S C1 IVF_S --batch 16384 --steps 3 | sed "s/^{/{\"variant\": \"$v\", /"
  QINCO_MLP_VARIANT=$v timeout 600 python scripts/bench_extra.py S --batch 1024 --steps 20 | sed "s/^{/{\"variant\": \"$v\", /"
done | tee $O/occ.jsonl | cut -c1-330
