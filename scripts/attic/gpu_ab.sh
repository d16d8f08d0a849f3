#!/bin/bash
# A/B of fused-MLP kernel variants (QINCO_MLP_VARIANT="P,VAR") + parity under the variant
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
V=${1:-"36,28"}
QINCO_MLP_VARIANT=$V timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu_variant.log 2>&1; tail -4 $O/pytest_gpu_variant.log
for v in "48,92" "$V" "48,92" "$V"; do
  echo "== C2 variant $v"; QINCO_MLP_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --batch 8192 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'])"
done
for v in "48,92" "$V"; do
  echo "== C1 variant $v"; QINCO_MLP_VARIANT=$v timeout 600 python scripts/bench_extra.py C1 --steps 3 2>/dev/null | grep encode | cut -c1-220
done
cd /tmp
QINCO_MLP_VARIANT=$V timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/prof_pmc_stall -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_stall.log 2>&1
