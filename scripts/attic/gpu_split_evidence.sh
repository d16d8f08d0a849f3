#!/bin/bash
# Evidence for the split-fp16 form: kernel trace + PMC passes (matrix-pipe busy, fp16 / fp32 MFMA ops, HBM bytes) of the C2 encode
# with QINCO_SPLIT_F16=1, and the traces of qinco2-S and C1.  Outputs -> gpurun_out/r02_split_*.
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
export QINCO_SPLIT_F16=1
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r02_$name; find $O/prof_$name -name '*.db' -delete
}
X="python $R/scripts/bench_extra.py"
prof split_c2_trace --kernel-trace --stats -d $O/prof_split_c2_trace -o t -- $X C2 --batch 16384 --steps 3
prof split_c2_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_split_c2_pmc_mfma -o t -- $X C2 --batch 8192 --steps 2
prof split_c2_pmc_fetch --pmc FETCH_SIZE --kernel-trace -d $O/prof_split_c2_pmc_fetch -o t -- $X C2 --batch 8192 --steps 2
prof split_c2_pmc_write --pmc WRITE_SIZE --kernel-trace -d $O/prof_split_c2_pmc_write -o t -- $X C2 --batch 8192 --steps 2
prof split_S_trace --kernel-trace --stats -d $O/prof_split_S_trace -o t -- $X S --batch 16384 --steps 3
prof split_C1_trace --kernel-trace --stats -d $O/prof_split_C1_trace -o t -- $X C1 --batch 16384 --steps 3
cd $R
for f in $O/r02_split_*_kernel_stats.csv; do echo == $f; head -6 $f | cut -c1-160; done
grep -h mlp_split $O/r02_split_c2_pmc_*_counters.csv | cut -c1-300 | head -12
