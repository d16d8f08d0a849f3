#!/bin/bash
# f3 top-k: throughput, per-kernel time and PMC counters (rocprofv3 -> CSV summaries in gpurun_out/)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python scripts/bench_extra.py knn 2>&1 | tail -4
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_knn -o knn -- python $R/scripts/bench_extra.py knn > $O/prof_knn.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_knn_pmc -o pmc -- python $R/scripts/bench_extra.py knn > $O/prof_knn_pmc.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $O/prof_knn_pmc2 -o pmc -- python $R/scripts/bench_extra.py knn > $O/prof_knn_pmc2.log 2>&1
cd $R
python scripts/rocpd_summary.py $O/prof_knn/knn_results.db $O/knn && grep -v "at::native" $O/knn_kernel_stats.csv | head -8
python scripts/rocpd_summary.py $O/prof_knn_pmc/pmc_results.db $O/knn_pmc && grep -v "at::native" $O/knn_pmc_counters.csv | cut -c1-400
python scripts/rocpd_summary.py $O/prof_knn_pmc2/pmc_results.db $O/knn_pmc2 && grep -v "at::native" $O/knn_pmc2_counters.csv | cut -c1-400
