#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for m in 0 1; do
  n=s1024_split$m
  QINCO_SPLIT_F16=$m timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o t -- python $R/scripts/bench_extra.py S --batch 1024 --steps 40 > $O/$n.log 2>&1
  db=$(find $O/prof_$n -name '*.db' | head -1); python $R/scripts/rocpd_summary.py $db $O/$n; find $O/prof_$n -name '*.db' -delete
  grep -v "^kernel" $O/${n}_by_grid.csv | sort -t, -k5 -n -r | head -9 | cut -c1-60,120-200
  grep vectors_per_s $O/$n.log | cut -c1-140
done
