#!/bin/bash
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_ivf -o pmc -- python $R/scripts/bench_extra.py IVF_S --beams 8 --steps 1 > $O/prof_ivf.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d $O/prof_ivf2 -o pmc -- python $R/scripts/bench_extra.py IVF_S --beams 8 --steps 1 > $O/prof_ivf2.log 2>&1
cd $R
timeout 300 python bench.py --batch 1024 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-200
