#!/bin/bash
# round 3: where a qinco2-S encode call spends its time at the reference's batch of 1024 and at 16384 (kernel trace by grid,
# plus the share of the wall clock that no kernel covers)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
for b in 1024 16384; do
  n=r03_S_b$b
  steps=$((40960 / b + 2))
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o t -- python $R/scripts/bench_extra.py S --batch $b --steps $steps > $O/$n.log 2>&1
  db=$(find $O/prof_$n -name '*.db' | head -1)
  python $R/scripts/rocpd_summary.py $db $O/$n
  python - $db <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
rows = cur.execute("select start, end, name from kernels order by start").fetchall()
# the encode calls: from the first normalize_kernel of the last third of the trace
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
small = [g for g in gaps if g < 200_000]
print(f"kernels {len(rows)}  busy {busy/1e6:.2f} ms  span {span/1e6:.2f} ms  median gap {sorted(gaps)[len(gaps)//2]/1e3:.2f} us  sum of gaps < 200 us: {sum(small)/1e6:.2f} ms")
PY
  find $O/prof_$n -name '*.db' -delete
  grep -v "^kernel" $O/${n}_by_grid.csv | sort -t, -k5 -n -r | head -12 | cut -c1-70,120-220
  grep vectors_per_s $O/$n.log | cut -c1-160
done
