#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
for st in 0 3 6 9 12; do
  QINCO_STAGGER=$st timeout 300 python scripts/bench_extra.py S C1 --batch 16384 --steps 3 | grep encode | sed "s/^{/{\"stagger\": $st, /" | cut -c1-200
done
timeout 300 python scripts/bench_extra.py S C1 --batch 16384 --steps 3 | grep encode | sed "s/^{/{\"stagger\": \"default\", /" | cut -c1-200
