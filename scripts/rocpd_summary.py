#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2 rocpd SQLite output) into small CSVs for profiles/.
   python scripts/rocpd_summary.py <results.db> <out_prefix>
Writes <out_prefix>_kernel_stats.csv (the --stats table) and, if counters were collected,
<out_prefix>_counters.csv (per kernel: mean duration and the per-dispatch mean of every counter, summed over
its dimensions)."""
import csv
import sqlite3
import sys
from collections import defaultdict


def main(db, prefix):
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, tot, avg, pct in rows:
            w.writerow([name, calls, f"{tot:.3f}", f"{avg:.3f}", f"{pct:.4f}"])
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "kernels" in tables:   # the same kernel at different launch sizes: one line per (kernel, grid)
        rows = cur.execute("select name, grid_x, grid_y, count(*), avg(duration), min(duration), max(duration) from kernels "
                           "group by name, grid_x, grid_y order by sum(duration) desc").fetchall()
        with open(prefix + "_by_grid.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_x_threads", "grid_y", "calls", "avg_us", "min_us", "max_us"])
            for name, gx, gy, n, avg, mn, mx in rows:
                w.writerow([name, gx, gy, n, f"{avg / 1e3:.3f}", f"{mn / 1e3:.3f}", f"{mx / 1e3:.3f}"])
    if "counters_collection" not in tables:
        return
    acc = defaultdict(lambda: defaultdict(float))     # (kernel, dispatch) -> counter -> summed value
    dur = {}
    for disp, kname, cname, val, start, end in cur.execute(
            "select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection"):
        acc[(kname, disp)][cname] += val
        dur[(kname, disp)] = (end - start) / 1e3
    per_kernel = defaultdict(list)
    for (kname, disp), c in acc.items():
        per_kernel[kname].append((dur[(kname, disp)], c))
    counters = sorted({c for v in acc.values() for c in v})
    with open(prefix + "_counters.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "avg_us"] + counters)
        for kname, lst in sorted(per_kernel.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
            n = len(lst)
            w.writerow([kname, n, f"{sum(d for d, _ in lst) / n:.3f}"] +
                       [f"{sum(c.get(cn, 0.0) for _, c in lst) / n:.1f}" for cn in counters])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
