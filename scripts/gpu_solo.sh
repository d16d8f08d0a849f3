#!/bin/bash
# the two-workgroups-per-CU instances with ONE workgroup per CU (experiment build: dummy dynamic LDS): every phase of a tile alone on its SIMD
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
for pad in 64 0; do for v in 380 4476; do
  echo "== pad $pad KiB variant $v"
  QINCO_RING_PAD_KIB=$pad QINCO_VARIANT=$v QINCO_HIP_LIB=scripts/exp_libs/lib_timeline.so python scripts/exp_timeline.py S 16384 2>&1 | grep '"encode' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v) if isinstance(v,float) else v) for k,v in d.items() if k not in ('unit','start_time_quantiles','launch','workload','makespan','selep_keys_published')})"
done; done | tee $O/solo_timeline.log
