#!/bin/bash
# round-2 session 2: selection rewrite (tests + kernel trace), ring checker, cpu baseline sweep
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
timeout 600 scripts/ubench/ring_check > $O/ring_check.log 2>&1; grep -c "" $O/ring_check.log; grep -v "^    wg" $O/ring_check.log | awk '{print}' | head -80
timeout 600 python scripts/bench_extra.py S IVF_S C2 --batch 16384 --steps 3 > $O/bench_extra_s2.jsonl 2> $O/bench_extra_s2.err; cat $O/bench_extra_s2.jsonl
timeout 600 python scripts/bench_extra.py S --batch 1024 --steps 20 >> $O/bench_extra_s2.jsonl 2>> $O/bench_extra_s2.err; tail -2 $O/bench_extra_s2.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_S -o trace -- python $R/scripts/bench_extra.py S IVF_S --batch 16384 --steps 3 > $O/prof_S.log 2>&1
cd $R
DB=$(find $O/prof_S -name '*.db' | head -1); python scripts/rocpd_summary.py $DB $O/s2_S && cat $O/s2_S_kernel_stats.csv; find $O/prof_S -name '*.db' -delete
timeout 900 python scripts/cpu_baseline_sweep.py C2 > $O/cpu_sweep.jsonl 2> $O/cpu_sweep.err; cat $O/cpu_sweep.jsonl
