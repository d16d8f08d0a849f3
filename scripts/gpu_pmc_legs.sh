#!/bin/bash
# One rocprofv3 --pmc pass per bench leg: the MFMA FLOPs the hardware counted (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512) against the count the
# library reports for the same calls (qinco_profile_read2, bench.py's roofline.frac).  usage: gpu_pmc_legs.sh "<workload> <mode> <rows>" ...
# -> gpurun_out/<round>_pmc_<wl>_<mode>_<rows>_counters.csv + one summary line per leg in gpurun_out/${ROUND:-r05}_pmc_legs.jsonl
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
: > $O/${ROUND:-r05}_pmc_legs.jsonl
for cfg in "$@"; do
  set -- $cfg
  n=${ROUND:-r05}_pmc_$1_$2_$3
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_$n -o t -- python $R/scripts/prof_calls.py $1 $2 $3 4 > $O/$n.log 2>&1
  db=$(find $O/prof_$n -name '*.db' | head -1); python $R/scripts/rocpd_summary.py $db $O/$n; rm -rf $O/prof_$n
  python - $O/${n}_counters.csv $O/$n.log "$cfg" >> $O/${ROUND:-r05}_pmc_legs.jsonl <<PY
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lib = [json.loads(l[8:]) for l in open(sys.argv[2]) if l.startswith("LIBRARY ")][0]
calls = lib["calls_counted"] + 1                      # (+ the warm-up call: the counters see every dispatch)
mops = sum(float(r["SQ_INSTS_VALU_MFMA_MOPS_F32"]) * int(r["dispatches"]) for r in rows if "mlp" in r["kernel"] or "xproj" in r["kernel"])
pmc_flops_per_call = mops * 512.0 / calls
top = max(rows, key=lambda r: float(r["avg_us"]) * int(r["dispatches"]))
print(json.dumps({"leg": sys.argv[3], "pmc_mfma_flops_per_call": pmc_flops_per_call, "library_executed_flops_per_call": lib["executed_flops_per_call"],
                  "library_over_pmc": lib["executed_flops_per_call"] / pmc_flops_per_call if pmc_flops_per_call else None,
                  "algorithmic_flops_per_call": lib["algorithmic_flops_per_call"], "dominant_kernel": top["kernel"][:80],
                  "dominant_kernel_avg_us_under_pmc": float(top["avg_us"])}))
PY
  tail -1 $O/${ROUND:-r05}_pmc_legs.jsonl
done
