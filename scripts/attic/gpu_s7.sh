#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 scripts/ubench/ring_check 3000 12 2>&1 | grep -v "^    wg" | grep "mfma 4" | awk '{print $1,$2,$3,$4,"|",$(NF-20),$(NF-19),$(NF-18),"|",$(NF-8),$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3),$(NF-2),$(NF-1),$NF}' | sort | uniq -c | head -40
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for v in "" "48,124"; do
  QINCO_MLP_VARIANT=$v timeout 600 python scripts/bench_extra.py S_d96 S_d768 --batch 16384 --steps 3 | sed "s/^{/{\"variant\": \"$v\", /"
done | tee $O/occ2.jsonl | cut -c1-250
