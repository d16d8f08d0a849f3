R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp; cd /tmp
QINCO_HIP_LIB=$R/scripts/exp_libs/$1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_exp -o t -- python $R/scripts/bench_extra.py knn > $O/exp_knn.log 2>&1
db=$(find $O/prof_exp -name '*.db' | head -1); [ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $O/exp_knn; rm -rf $O/prof_exp
grep "knn_table" $O/exp_knn_by_grid.csv | cut -c1-60,150-300
