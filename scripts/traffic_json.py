#!/usr/bin/env python
"""profiles/rNN_c2_traffic.json from the three PMC passes of the bench command (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE
are in KiB, FETCH_SIZE counts half of the bytes on gfx950 -> x2).   python scripts/traffic_json.py <prefix> <rows_per_launch>
reads <prefix>_pmc_{fetch,write,mfma}_counters.csv, writes <prefix>_traffic.json."""
import csv
import json
import sys

prefix, rows = sys.argv[1], int(sys.argv[2])


def row(kind, kernel="mlp_kernel<"):
    with open(f"{prefix}_pmc_{kind}_counters.csv") as f:
        for r in csv.DictReader(f):
            if kernel in r["kernel"]:
                return r
    raise SystemExit(f"no {kernel} in {kind}")


f, w, m = row("fetch"), row("write"), row("mfma")
fetch = float(f["FETCH_SIZE"]) * 1024 * 2
write = float(w["WRITE_SIZE"]) * 1024
us = float(m["avg_us"])
flops = float(m["SQ_INSTS_VALU_MFMA_MOPS_F32"]) * 512
out = {"kernel": m["kernel"], "rows_per_launch": rows, "FETCH_SIZE_kb": float(f["FETCH_SIZE"]), "WRITE_SIZE_kb": float(w["WRITE_SIZE"]),
       "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "bytes_per_row": (fetch + write) / rows, "avg_launch_us": us,
       "mfma_flops_executed_per_launch": flops, "mfma_executed_tflops": flops / us / 1e6,
       "mfma_executed_frac_of_peak": flops / us / 1e6 / 157.3,
       "clock_GHz": float(m["GRBM_GUI_ACTIVE"]) / 8 / us / 1e3,
       "note": "L2<->fabric bytes per launch of the fused-MLP kernel from separate rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 "
               "correction per MI355X_MICROARCH.md, WRITE_SIZE); Infinity-Cache hits are included, so this is an upper bound on HBM "
               "traffic.  SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 = FLOPs the matrix pipe executed; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration."}
json.dump(out, open(f"{prefix}_traffic.json", "w"), indent=1)
print(json.dumps(out))
