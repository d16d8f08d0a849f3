#!/bin/bash
# counter passes over a qinco2-S encode: where the waves of the short-MLP kernel wait (issue stalls, LDS, vector memory path)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
run() {
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d $O/prof_$name -o t -- python $R/scripts/prof_calls.py S encode 16384 3 > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1); python $R/scripts/rocpd_summary.py $db $O/r04_$name; rm -rf $O/prof_$name
  grep -E "^kernel|mlp_kernel" $O/r04_${name}_counters.csv | cut -c1-400
}
run S_pmc_a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run S_pmc_b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD
# (a third pass with TA_* / TCP_* names hung rocprofv3 until its timeout on this image: not repeated)
