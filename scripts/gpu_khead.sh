#!/bin/bash
# KHEAD (VAR 4476) against the production two-workgroups-per-CU instance (380): throughput + identity, then per-SIMD schedules
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python scripts/exp_khead_ab.py S --beam 8 8 --beam 20 8 --beam 1 1 --beam 32 4 --beam 64 2 > $O/khead_ab.log 2>&1; tail -n 14 $O/khead_ab.log
python scripts/exp_khead_ab.py C1 >> $O/khead_ab.log 2>&1; tail -n 6 $O/khead_ab.log
if [ -f scripts/exp_libs/lib_timeline.so ]; then
for v in 380 4476; do
  QINCO_SCHEDULE_OUT=$O/sched_$v.npy QINCO_VARIANT=$v QINCO_HIP_LIB=scripts/exp_libs/lib_timeline.so python scripts/exp_cu_schedule.py S 16384 2>&1 | tail -n 1
  QINCO_VARIANT=$v QINCO_HIP_LIB=scripts/exp_libs/lib_timeline.so python scripts/exp_timeline.py S 16384 2>&1 | grep '"encode' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v) if isinstance(v,float) else v) for k,v in d.items() if k not in ('unit','start_time_quantiles','launch','workload')})"
done | tee $O/khead_timeline.log
fi
