#!/bin/bash
# Round-5 final evidence on the final build: GPU suite, the driver line, kernel trace + PMC passes of the same command, one PMC pass per
# short-MLP leg, the small-call traces, the selection experiment, sweeps, the multi-rank bench on the one GPU (gloo).
# Outputs -> gpurun_out/r05f_*  (copy what is to be judged into profiles/).
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; export ROUND=r05f
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r05f_pytest_gpu.txt 2>&1; tail -n 3 $O/r05f_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 1500 python bench.py > $O/r05f_bench_c2_n1.json 2> $O/r05f_bench.err; head -c 300 $O/r05f_bench_c2_n1.json; echo
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r05f_$name; rm -rf $O/prof_$name
}
B="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-legs"
prof c2_trace --kernel-trace --stats -d $O/prof_c2_trace -o t -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs
prof c2_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_c2_pmc_mfma -o t -- $B --batch 8192
prof c2_pmc_fetch --pmc FETCH_SIZE --kernel-trace -d $O/prof_c2_pmc_fetch -o t -- $B --batch 8192
prof c2_pmc_write --pmc WRITE_SIZE --kernel-trace -d $O/prof_c2_pmc_write -o t -- $B --batch 8192
python $R/scripts/traffic_json.py $O/r05f_c2 1048576
cd $R
bash scripts/gpu_prof_calls.sh "S encode 16384" "S encode 1024" "IVF_S encode 16384" > $O/r05f_prof_calls.log 2>&1; grep -E "^===|vectors/s|dist_topk|presel|mlp_kernel|ivf_f16" $O/r05f_prof_calls.log | cut -c1-150
bash scripts/gpu_pmc_legs.sh "S encode 16384" "C1 encode 16384" "C2 encode 16384" > $O/r05f_pmc_legs.log 2>&1; cat $O/r05f_pmc_legs.jsonl | cut -c1-260
COOP=0 TS=1,2,8,12,16,17,32 python scripts/exp_pair_select.py 2>&1 | grep -v amdgpu.ids > $O/r05f_exp_pair_select.log; cut -c1-170 $O/r05f_exp_pair_select.log
timeout 900 python tests/sweeps/gpu_fuzz_inputs.py --seed 9 --out $O/r05f_fuzz_inputs_seed9.jsonl > $O/r05f_fuzz_inputs.log 2>&1; tail -n 2 $O/r05f_fuzz_inputs.log
timeout 1200 python tests/sweeps/gpu_fuzz_geometry.py --seed 43 --count 16 --out $O/r05f_fuzz_geometry_seed43.jsonl > $O/r05f_fuzz_geometry.log 2>&1; tail -n 2 $O/r05f_fuzz_geometry.log
timeout 600 python bench.py --gpus 8 --backend gloo --steps 2 --warmup 1 --batch 4096 > $O/r05f_bench_c2_n8_gloo_shared_gpu_weak.json 2> $O/r05f_n8.err; head -c 200 $O/r05f_bench_c2_n8_gloo_shared_gpu_weak.json; echo
timeout 300 python bench.py --gpus 8 --backend gloo --dry-rccl > $O/r05f_dry_rccl_n8_gloo_shared_gpu.json 2>> $O/r05f_n8.err; head -c 300 $O/r05f_dry_rccl_n8_gloo_shared_gpu.json; echo
ls $O/r05f_* | wc -l
