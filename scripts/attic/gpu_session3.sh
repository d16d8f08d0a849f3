#!/bin/bash
# session 3: 32-row ring kernels without the exclusive-LDS padding: throughput and a bitwise stress across variants
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for pad in 36 0; do
  QINCO_RING_PAD_KIB=$pad timeout 600 python scripts/bench_extra.py S M C1 IVF_S --batch 16384 --steps 3 | sed "s/^{/{\"pad_kib\": $pad, /" >> $O/s3_pad.jsonl
  QINCO_RING_PAD_KIB=$pad timeout 600 python scripts/bench_extra.py S --batch 1024 --steps 20 | sed "s/^{/{\"pad_kib\": $pad, /" >> $O/s3_pad.jsonl
done
cat $O/s3_pad.jsonl
QINCO_RING_PAD_KIB=0 timeout 900 python scripts/gpu_stress.py S C1 M --n 32768 > $O/s3_stress.log 2>&1; tail -12 $O/s3_stress.log
