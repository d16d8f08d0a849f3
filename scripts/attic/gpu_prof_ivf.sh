#!/bin/bash
# IVF step-0 kernel: kernel-trace durations (no counters), then PMC passes
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_ivf0 -o trace -- python $R/scripts/bench_extra.py IVF_S --beams 8 --steps 2 > $O/prof_ivf0.log 2>&1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_ivf -o pmc -- python $R/scripts/bench_extra.py IVF_S --beams 8 --steps 1 > $O/prof_ivf.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d $O/prof_ivf2 -o pmc -- python $R/scripts/bench_extra.py IVF_S --beams 8 --steps 1 > $O/prof_ivf2.log 2>&1
cd $R
python scripts/rocpd_summary.py $O/prof_ivf0/trace_results.db $O/ivf_trace && head -8 $O/ivf_trace_kernel_stats.csv | cut -c1-200
python scripts/rocpd_summary.py $O/prof_ivf/pmc_results.db $O/ivf_pmc && grep "ivf_\|^kernel" $O/ivf_pmc_counters.csv | cut -c1-400
python scripts/rocpd_summary.py $O/prof_ivf2/pmc_results.db $O/ivf_pmc2 && grep "ivf_\|^kernel" $O/ivf_pmc2_counters.csv | cut -c1-400
