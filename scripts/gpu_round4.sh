#!/bin/bash
# Round-4 evidence: the GPU suite, the driver line, kernel trace + PMC passes of the same command, traces of the small-launch calls.
# Outputs -> gpurun_out/r04_*  (copy what is to be judged into profiles/).
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/r04_pytest_gpu.txt 2>&1; tail -3 $O/r04_pytest_gpu.txt
timeout 1500 python bench.py > $O/r04_bench_c2_n1.json 2> $O/r04_bench.err; head -c 400 $O/r04_bench_c2_n1.json; echo
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r04_$name; rm -rf $O/prof_$name
}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-legs"
prof c2_trace --kernel-trace --stats -d $O/prof_c2_trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs
prof c2_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_c2_pmc_mfma -o t -- $B --batch 8192
prof c2_pmc_fetch --pmc FETCH_SIZE --kernel-trace -d $O/prof_c2_pmc_fetch -o t -- $B --batch 8192
prof c2_pmc_write --pmc WRITE_SIZE --kernel-trace -d $O/prof_c2_pmc_write -o t -- $B --batch 8192
python $R/scripts/traffic_json.py $O/r04_c2 1048576
cd $R
scripts/gpu_prof_calls.sh "C2 decode 12288" "C2 decode 1024" "C1 decode 12288" "S decode 12288" "S decode 1024" "C2 encode1 1024" "S encode 1024" > $O/r04_prof_calls.log 2>&1
tail -40 $O/r04_prof_calls.log
ls $O/r04_* | wc -l
