#!/bin/bash
# rocprofv3 kernel trace of small calls: gpu_prof_calls.sh "<workload> <mode> <rows>" ...  -> gpurun_out/<round>_<wl>_<mode>_<rows>_{kernel_stats,by_grid}.csv
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for cfg in "$@"; do
  set -- $cfg
  n=${ROUND:-r05}_$1_$2_$3
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o t -- python $R/scripts/prof_calls.py $1 $2 $3 20 > $O/$n.log 2>&1
  db=$(find $O/prof_$n -name '*.db' | head -1); python $R/scripts/rocpd_summary.py $db $O/$n; rm -rf $O/prof_$n
  echo "=== $cfg"; grep "vectors/s" $O/$n.log
  python - $O/${n}_by_grid.csv <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["avg_us"])*int(r["calls"]) for r in rows)
for r in rows[:10]:
    t=float(r["avg_us"])*int(r["calls"])
    print(f'{r["kernel"][:70]:70s} grid {r["grid_x_threads"]:>9s} calls {r["calls"]:>4s} avg {float(r["avg_us"]):9.1f} us  {100*t/tot:5.1f} %')
print("total kernel time per call (us):", tot/21)
PY
done
