#!/bin/bash
# kernel-time breakdown (rocprofv3 --kernel-trace --stats) of bench_extra workloads
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for wl in "$@"; do
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_wl_$wl -o trace -- python $R/scripts/bench_extra.py $wl --beams 8 --steps 2 > $O/prof_wl_$wl.log 2>&1
  grep workload $O/prof_wl_$wl.log
done
