#!/bin/bash
# kernel traces of the IVF-qinco2-S encode (fp32 path and split form) after the sampled pass A
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for m in 0 1; do
  n=r02_ivfS_sampled_split$m
  QINCO_SPLIT_F16=$m timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o t -- python $R/scripts/bench_extra.py IVF_S --batch 16384 --steps 3 > $O/$n.log 2>&1
  db=$(find $O/prof_$n -name '*.db' | head -1); python $R/scripts/rocpd_summary.py $db $O/$n; find $O/prof_$n -name '*.db' -delete
  head -8 $O/${n}_kernel_stats.csv | cut -c1-150
done
