#!/bin/bash
# Round-2 evidence: bench line, kernel trace + PMC passes of the same command, extra workloads.  Outputs -> gpurun_out/r02_*.
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py > $O/r02_bench_c2_n1.json 2> $O/r02_bench.err; cat $O/r02_bench_c2_n1.json
timeout 600 python bench.py --gpus 2 --backend gloo --steps 3 --batch 8192 > $O/r02_bench_c2_n2_gloo_shared_gpu.json 2>> $O/r02_bench.err; cat $O/r02_bench_c2_n2_gloo_shared_gpu.json
timeout 1200 python scripts/bench_extra.py C1 C2 C3 C4 S M IVF_S IVF_L Q1_768 S_d96 S_d768 --batch 16384 --steps 3 > $O/r02_bench_extra.jsonl 2> $O/r02_bench_extra.err
timeout 600 python scripts/bench_extra.py C1 C2 --beams 1 8 --batch 16384 --steps 3 >> $O/r02_bench_extra.jsonl 2>> $O/r02_bench_extra.err
timeout 600 python scripts/bench_extra.py S C2 --batch 1024 --steps 20 >> $O/r02_bench_extra.jsonl 2>> $O/r02_bench_extra.err
cat $O/r02_bench_extra.jsonl | cut -c1-260
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r02_$name; find $O/prof_$name -name '*.db' -delete
}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras"
prof c2_trace --kernel-trace --stats -d $O/prof_c2_trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline
prof c2_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_c2_pmc_mfma -o t -- $B --batch 8192
prof c2_pmc_fetch --pmc FETCH_SIZE --kernel-trace -d $O/prof_c2_pmc_fetch -o t -- $B --batch 8192
prof c2_pmc_write --pmc WRITE_SIZE --kernel-trace -d $O/prof_c2_pmc_write -o t -- $B --batch 8192
prof S_trace --kernel-trace --stats -d $O/prof_S_trace -o t -- python $R/scripts/bench_extra.py S --batch 16384 --steps 3
prof ivfS_trace --kernel-trace --stats -d $O/prof_ivfS_trace -o t -- python $R/scripts/bench_extra.py IVF_S --batch 16384 --steps 3
prof S_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_S_pmc_mfma -o t -- python $R/scripts/bench_extra.py S --batch 16384 --steps 2
cd $R
ls $O/r02_* | head -40
