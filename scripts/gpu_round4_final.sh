#!/bin/bash
# Round-4 final evidence on the final build: GPU suite, stress of KHEAD, the driver line, kernel trace + PMC passes of the same command,
# one PMC pass per short-MLP leg.  Outputs -> gpurun_out/r04f_*  (copy what is to be judged into profiles/).
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/r04f_pytest_gpu.txt 2>&1; tail -n 3 $O/r04f_pytest_gpu.txt
timeout 600 python scripts/stress_khead.py 2>&1 | grep -v amdgpu.ids > $O/r04f_stress_khead.jsonl; cat $O/r04f_stress_khead.jsonl
timeout 1500 python bench.py > $O/r04f_bench_c2_n1.json 2> $O/r04f_bench.err; head -c 300 $O/r04f_bench_c2_n1.json; echo
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r04f_$name; rm -rf $O/prof_$name
}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-legs"
prof c2_trace --kernel-trace --stats -d $O/prof_c2_trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs
prof c2_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_c2_pmc_mfma -o t -- $B --batch 8192
prof c2_pmc_fetch --pmc FETCH_SIZE --kernel-trace -d $O/prof_c2_pmc_fetch -o t -- $B --batch 8192
prof c2_pmc_write --pmc WRITE_SIZE --kernel-trace -d $O/prof_c2_pmc_write -o t -- $B --batch 8192
python $R/scripts/traffic_json.py $O/r04f_c2 1048576
prof S_trace --kernel-trace --stats -d $O/prof_S_trace -o t -- python $R/scripts/prof_calls.py S encode 16384 6
cd $R
bash scripts/gpu_pmc_legs.sh "S encode 16384" "C1 encode 16384" "C2 encode 16384" > $O/r04f_pmc_legs.log 2>&1; cp $O/r04_pmc_legs.jsonl $O/r04f_pmc_legs.jsonl; cat $O/r04f_pmc_legs.jsonl | cut -c1-260
ls $O/r04f_* | wc -l
