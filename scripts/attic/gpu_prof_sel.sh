#!/bin/bash
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sel -o trace -- python $R/scripts/bench_select.py 16384 > $O/prof_sel.log 2>&1
cd $R
DB=$(find $O/prof_sel -name '*.db' | head -1); python scripts/rocpd_summary.py $DB $O/${1:-sel} && grep -E "dist_topk|beam_select|xproj" $O/${1:-sel}_by_grid.csv; find $O/prof_sel -name '*.db' -delete
