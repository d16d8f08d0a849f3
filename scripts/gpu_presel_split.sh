#!/bin/bash
# Time of the pre-selection kernels as a function of T (= A): one rocprofv3 kernel trace per A, large (16384 x 8 groups) and small (1024 x 8) launches.
# -> gpurun_out/${ROUND}_presel_T<A>_<n>_by_grid.csv ; prints the table kernels' lines
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for n in 16384 1024; do
for A in ${AS:-1 8 16 32}; do
  t=${ROUND:-r05}_presel_T${A}_$n
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$t -o t -- python $R/scripts/bench_select.py $n $A > $O/$t.log 2>&1
  db=$(find $O/prof_$t -name '*.db' | head -1); python $R/scripts/rocpd_summary.py $db $O/$t > /dev/null; rm -rf $O/prof_$t
  echo "== T=$A n=$n"; grep -E "dist_topk|presel_xproj" $O/${t}_by_grid.csv | cut -c1-60,150-
done; done
