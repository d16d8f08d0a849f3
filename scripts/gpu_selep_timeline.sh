#!/bin/bash
# Timeline of a qinco2-S encode step with and without the epilogue selection (experiment build with cycle stamps)
python scripts/build_exp_lib.py timeline "-DQINCO_TIMELINE" 128,128,256,48,380 128,128,256,48,2428 > /dev/null 2>&1 || exit 1
for v in 0 1; do
  echo "== QINCO_NO_SELEP=$v"
  if [ $v = 1 ]; then export QINCO_NO_SELEP=1; else unset QINCO_NO_SELEP; fi
  QINCO_HIP_LIB=scripts/exp_libs/lib_timeline.so python scripts/exp_timeline.py S 16384 2>&1 | grep '"encode' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v) if isinstance(v,float) else v) for k,v in d.items() if k not in ('unit','start_time_quantiles','launch','workload')})"
done
