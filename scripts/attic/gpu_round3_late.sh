#!/bin/bash
# Round-3, late additions: the whole GPU suite on the final build, the driver line again, QINCo1 at D = 768 with the folded 16-row
# form (record + kernel trace + matrix-pipe counters), and the four geometry-sweep seeds on the final build (the on-demand planner
# now sends wide shapes to the folded 16-row form).  Outputs -> gpurun_out/r03_*  (copy what is to be judged into profiles/).
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/r03_late_pytest.txt; cat $O/r03_late_pytest.txt
timeout 1500 python bench.py > $O/r03_late_bench_c2_n1.json 2> $O/r03_late_bench.err; head -c 400 $O/r03_late_bench_c2_n1.json
timeout 600 python scripts/bench_extra.py Q1_768 --batch 16384 --steps 3 > $O/r03_late_bench_extra_q1_768.jsonl 2> $O/r03_late_bench_extra.err
cut -c1-260 $O/r03_late_bench_extra_q1_768.jsonl
timeout 300 python scripts/exp_fold16.py 2>&1 | grep -v amdgpu.ids > $O/r03_exp_fold16.log; cat $O/r03_exp_fold16.log
for s in 7 11 13 17; do
  c=48; [ $s = 7 ] && c=32; [ $s = 11 ] && c=40
  timeout 900 python tests/sweeps/gpu_fuzz_geometry.py --seed $s --count $c --out $O/r03_fuzz_geometry_seed$s.jsonl 2>&1 | grep -v '"ok": true' | tail -4
done
cd /tmp
prof() {  # name, rocprof args..., then the command after --
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.log 2>&1
  local db=$(find $O/prof_$name -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/r03_$name; find $O/prof_$name -name '*.db' -delete
}
prof q1_768_trace --kernel-trace --stats -d $O/prof_q1_768_trace -o t -- python $R/scripts/bench_extra.py Q1_768 --batch 16384 --steps 2
prof q1_768_pmc_mfma --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_q1_768_pmc_mfma -o t -- python $R/scripts/bench_extra.py Q1_768 --batch 8192 --steps 1
cd $R
ls $O/r03_late* $O/r03_q1* $O/r03_fuzz* $O/r03_exp_fold16.log
