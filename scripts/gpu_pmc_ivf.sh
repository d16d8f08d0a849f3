#!/bin/bash
# Counter-based look at the IVF coarse assignment (ivf_f16_kernel passes A / B) inside an IVF-qinco2-S encode: matrix-pipe busy share,
# wave wait / active split, fp16 MFMA ops -> gpurun_out/r04_ivf_pmc{1,2}_counters.csv
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
run() {
  local name=$1; shift
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace -d $O/prof_$name -o t -- python $R/scripts/prof_calls.py IVF_S encode 16384 3 > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1); python $R/scripts/rocpd_summary.py $db $O/r04_$name; rm -rf $O/prof_$name
  grep -E "^kernel|ivf_f16|ivf_exact|ivf_assign" $O/r04_${name}_counters.csv | cut -c1-400
}
run ivf_pmc1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
run ivf_pmc2 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
