#!/bin/bash
# Round-3 evidence: the driver line, kernel trace + PMC passes of the same command, every workload, small-batch runs.
# Outputs -> gpurun_out/r03_*  (copy what is to be judged into profiles/).
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python bench.py > $O/r03_bench_c2_n1.json 2> $O/r03_bench.err; head -c 600 $O/r03_bench_c2_n1.json
timeout 1500 python scripts/bench_extra.py C1 C2 C3 C4 S M IVF_S IVF_L Q1_768 S_d96 S_d768 --batch 16384 --steps 3 > $O/r03_bench_extra.jsonl 2> $O/r03_bench_extra.err
timeout 600 python scripts/bench_extra.py S C2 C1 M --batch 1024 --steps 20 >> $O/r03_bench_extra.jsonl 2>> $O/r03_bench_extra.err
cut -c1-200 $O/r03_bench_extra.jsonl
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r03_$name; find $O/prof_$name -name '*.db' -delete
}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-legs"
prof c2_trace --kernel-trace --stats -d $O/prof_c2_trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs
prof c2_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_c2_pmc_mfma -o t -- $B --batch 8192
prof c2_pmc_fetch --pmc FETCH_SIZE --kernel-trace -d $O/prof_c2_pmc_fetch -o t -- $B --batch 8192
prof c2_pmc_write --pmc WRITE_SIZE --kernel-trace -d $O/prof_c2_pmc_write -o t -- $B --batch 8192
python $R/scripts/traffic_json.py $O/r03_c2 1048576
prof S_trace --kernel-trace --stats -d $O/prof_S_trace -o t -- python $R/scripts/bench_extra.py S --batch 16384 --steps 3
prof S1024_trace --kernel-trace --stats -d $O/prof_S1024_trace -o t -- python $R/scripts/bench_extra.py S --batch 1024 --steps 40
cd $R
ls $O/r03_* | head -60
