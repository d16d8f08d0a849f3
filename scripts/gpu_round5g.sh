#!/bin/bash
# Late round 5 (after the filtered small-db search): GPU suite, smoke, the driver line on the final build. Outputs -> gpurun_out/r05g_*
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r05g_pytest_gpu.txt 2>&1; tail -n 3 $O/r05g_pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
timeout 1500 python bench.py > $O/r05g_bench_c2_n1.json 2> $O/r05g_bench.err; head -c 400 $O/r05g_bench_c2_n1.json; echo
