#!/bin/bash
# Small-db search (f3): GPU tests, both forms timed, a rocprofv3 kernel trace and one PMC pass of the same command.
# Outputs -> gpurun_out/${ROUND}_knn_*   (copy what is to be judged into profiles/).
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; ROUND=${ROUND:-r05g}
timeout 600 python -m pytest tests/test_search_eval.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 2
timeout 300 python scripts/bench_extra.py knn 2>/dev/null > $O/${ROUND}_knn_bench.jsonl; cut -c1-230 $O/${ROUND}_knn_bench.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_knn_trace -o t -- python $R/scripts/bench_extra.py knn > $O/${ROUND}_knn_trace.log 2>&1
db=$(find $O/prof_knn_trace -name '*.db' | head -1); [ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $O/${ROUND}_knn_trace; rm -rf $O/prof_knn_trace
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_knn_pmc -o t -- python $R/scripts/bench_extra.py knn > $O/${ROUND}_knn_pmc.log 2>&1
db=$(find $O/prof_knn_pmc -name '*.db' | head -1); [ -n "$db" ] && python $R/scripts/rocpd_summary.py $db $O/${ROUND}_knn_pmc; rm -rf $O/prof_knn_pmc
ls $O/${ROUND}_knn_* < /dev/null
