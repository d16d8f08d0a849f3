#!/bin/bash
# kernel trace of the C2 encode with the split-fp16 FFN blocks (QINCO_SPLIT_F16=1 switches qinco_create)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
QINCO_SPLIT_F16=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_split -o trace -- python $R/scripts/bench_extra.py C2 --batch 16384 --steps 3 > $O/prof_split.log 2>&1
cd $R
DB=$(find $O/prof_split -name '*.db' | head -1); python scripts/rocpd_summary.py $DB $O/${1:-r02_split}_c2 && cat $O/${1:-r02_split}_c2_kernel_stats.csv; find $O/prof_split -name '*.db' -delete
tail -3 $O/prof_split.log
