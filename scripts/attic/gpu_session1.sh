#!/bin/bash
# round-2 session 1: new parity tests, bench line, co-residency experiment
set -x
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
timeout 900 python scripts/exp_coresidency.py > $O/exp_cores.jsonl 2> $O/exp_cores.err; cat $O/exp_cores.jsonl; tail -3 $O/exp_cores.err
