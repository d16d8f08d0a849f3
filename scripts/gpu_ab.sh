#!/bin/bash
# A/B of fused-MLP kernel variants (QINCO_MLP_VARIANT="P,VAR") + parity of the production instances.
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
for v in "8,0" "36,12" "8,8" "8,0" "36,12"; do
  echo "== C2 variant $v"; QINCO_MLP_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --batch 8192 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'])"
done
for v in "8,0" "8,8" "36,12"; do
  echo "== C1 variant $v"; QINCO_MLP_VARIANT=$v timeout 600 python scripts/bench_extra.py C1 --steps 3 2>/dev/null | grep encode
done
timeout 600 python scripts/bench_extra.py C3 C4 --beams 8 --steps 2 2>/dev/null
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $O/prof_pmc_stall -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_stall.log 2>&1
cd $R; ls $O
