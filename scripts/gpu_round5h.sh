#!/bin/bash
# Round 5, FINAL build (filtered small-db search with the LDS prefetch and canonical NaNs): GPU suite, smoke, the driver line,
# the small-db search timed in both forms + its kernel trace, the knn test file ten times in a row.  Outputs -> gpurun_out/r05h_*
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r05h_pytest_gpu.txt 2>&1; tail -n 3 $O/r05h_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 1500 python bench.py > $O/r05h_bench_c2_n1.json 2> $O/r05h_bench.err; head -c 300 $O/r05h_bench_c2_n1.json; echo
timeout 300 python scripts/bench_extra.py knn 2>/dev/null > $O/r05h_knn_bench.jsonl; cut -c1-200 $O/r05h_knn_bench.jsonl
for i in 1 2 3 4 5 6 7 8 9 10; do python -m pytest tests/test_search_eval.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed"; done > $O/r05h_knn_tests_x10.txt; sort $O/r05h_knn_tests_x10.txt | cut -c1-30 | uniq -c
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_knn_trace -o t -- python $R/scripts/bench_extra.py knn > $O/r05h_knn_trace.log 2>&1
db=$(find $O/prof_knn_trace -name '*.db' | head -1); [ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $O/r05h_knn_trace; rm -rf $O/prof_knn_trace
grep "knn_table_kernel<128, true>" $O/r05h_knn_trace_by_grid.csv < /dev/null | cut -c1-50,140-260
