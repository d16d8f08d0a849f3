#!/bin/bash
# PMC stall / MFMA counters for one bench_extra workload:  bash scripts/gpu_prof_wl_pmc.sh S
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_wl_pmc -o pmc -- python $R/scripts/bench_extra.py $1 --beams 8 --steps 1 --batch 8192 > $O/prof_wl_pmc.log 2>&1
cd $R
python scripts/rocpd_summary.py $O/prof_wl_pmc/pmc_results.db $O/wl_pmc && grep "mlp_kernel\|xproj\|^kernel" $O/wl_pmc_counters.csv | cut -c1-330
