#!/bin/bash
# Round-6 final evidence on the final build: GPU suite, smoke, the driver line (traffic measured in-run, rccl_world1), kernel trace + PMC
# passes of the same command, the C1 leg at BASELINE.md's protocol (10 000 CPU vectors), the world-of-one RCCL run, the small-db search
# (bench, trace, PMC), per-leg traces of the S family, sweeps, the multi-rank bench on the one GPU (gloo).
# Outputs -> gpurun_out/r06f_*  (copy what is to be judged into profiles/).
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; export ROUND=r06f
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r06f_pytest_gpu.txt 2>&1; tail -n 3 $O/r06f_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
( time timeout 1500 python bench.py > $O/r06f_bench_c2_n1.json 2> $O/r06f_bench.err ) 2>&1 | grep real; head -c 300 $O/r06f_bench_c2_n1.json; echo
timeout 1500 python bench.py --c1-cpu-vectors 10000 --no-extras --no-cpu-baseline --no-pmc --no-rccl-check > $O/r06f_bench_c1_cpu_10000.json 2> $O/r06f_bench_c1.err
python - <<PY
import json
r = json.load(open("$O/r06f_bench_c1_cpu_10000.json"))
print("c1:", json.dumps(r.get("c1"))[:700])
PY
timeout 300 python bench.py --gpus 1 --dry-rccl > $O/r06f_dry_rccl_world1.json 2>> $O/r06f_bench.err; head -c 400 $O/r06f_dry_rccl_world1.json; echo
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r06f_$name; rm -rf $O/prof_$name
}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-legs --no-pmc --no-rccl-check"
prof c2_trace --kernel-trace --stats -d $O/prof_c2_trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-pmc --no-rccl-check
prof c2_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_c2_pmc_mfma -o t -- $B --batch 8192
prof c2_pmc_fetch --pmc FETCH_SIZE --kernel-trace -d $O/prof_c2_pmc_fetch -o t -- $B --batch 8192
prof c2_pmc_write --pmc WRITE_SIZE --kernel-trace -d $O/prof_c2_pmc_write -o t -- $B --batch 8192
python $R/scripts/traffic_json.py $O/r06f_c2 1048576
cd $R
ROUND=r06f bash scripts/gpu_knn_prof.sh > $O/r06f_knn_prof.log 2>&1; cut -c1-200 $O/r06f_knn_bench.jsonl
bash scripts/gpu_prof_calls.sh "S encode 16384" "IVF_S encode 16384" > $O/r06f_prof_calls.log 2>&1; grep -E "^===|vectors/s|mlp_kernel|ivf_f16" $O/r06f_prof_calls.log | cut -c1-150
bash scripts/gpu_pmc_legs.sh "S encode 16384" "C1 encode 16384" "C2 encode 16384" > $O/r06f_pmc_legs.log 2>&1; cat $O/r06f_pmc_legs.jsonl | cut -c1-260
timeout 900 python tests/sweeps/gpu_fuzz_inputs.py --seed 11 --out $O/r06f_fuzz_inputs_seed11.jsonl > $O/r06f_fuzz_inputs.log 2>&1; tail -n 2 $O/r06f_fuzz_inputs.log
timeout 1200 python tests/sweeps/gpu_fuzz_geometry.py --seed 47 --count 16 --out $O/r06f_fuzz_geometry_seed47.jsonl > $O/r06f_fuzz_geometry.log 2>&1; tail -n 2 $O/r06f_fuzz_geometry.log
timeout 600 python tests/sweeps/gpu_fuzz_knn.py --seed 4 --out $O/r06f_fuzz_knn_seed4.jsonl > $O/r06f_fuzz_knn.log 2>&1; tail -n 1 $O/r06f_fuzz_knn.log
timeout 600 python tests/sweeps/gpu_fuzz_ivf.py --seed 6 --out $O/r06f_fuzz_ivf_seed6.jsonl > $O/r06f_fuzz_ivf.log 2>&1; tail -n 1 $O/r06f_fuzz_ivf.log
timeout 600 python bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 --batch 4096 > $O/r06f_bench_c2_n8_gloo_shared_gpu_weak.json 2> $O/r06f_n8.err; head -c 200 $O/r06f_bench_c2_n8_gloo_shared_gpu_weak.json; echo
ls $O/r06f_* | wc -l
