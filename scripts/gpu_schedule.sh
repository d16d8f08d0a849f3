#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
for lib in timeline timeline12; do for v in 380 4476; do
  echo "== $lib variant $v"
  QINCO_SCHEDULE_OUT=$O/sched_${lib}_$v.npy QINCO_VARIANT=$v QINCO_HIP_LIB=scripts/exp_libs/lib_$lib.so python scripts/exp_cu_schedule.py S 16384 2>&1 | tail -n 1
done; done | tee $O/schedule.log
