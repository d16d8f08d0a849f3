#!/bin/bash
# kernel trace of the qinco2-S / IVF-qinco2-S encode (selection + xproj + mlp split)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "golden or selftest or fresh or instance" 2>&1 | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_S -o trace -- python $R/scripts/bench_extra.py S --batch 16384 --steps 3 > $O/prof_S.log 2>&1
cd $R
DB=$(find $O/prof_S -name '*.db' | head -1); python scripts/rocpd_summary.py $DB $O/${1:-s}_S && cat $O/${1:-s}_S_kernel_stats.csv; find $O/prof_S -name '*.db' -delete
tail -2 $O/prof_S.log
