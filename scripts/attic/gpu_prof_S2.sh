#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for tag in gp8 gp4; do
  if [ $tag = gp4 ]; then export QINCO_HIP_LIB=$R/scripts/exp_libs/lib_gp4.so; fi
  timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or selftest" 2>&1 | tail -1
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_S -o trace -- python $R/scripts/bench_extra.py S --batch 16384 --steps 3 > $O/prof_S.log 2>&1
  cd $R
  DB=$(find $O/prof_S -name '*.db' | head -1); python scripts/rocpd_summary.py $DB $O/s5_$tag && head -5 $O/s5_${tag}_kernel_stats.csv; find $O/prof_S -name '*.db' -delete
done
