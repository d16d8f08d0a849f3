#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace + PMC passes.  Outputs -> gpurun_out/.
set -x
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
timeout 900 python scripts/bench_extra.py C1 C2 C3 C4 --beams 1 8 > $O/bench_extra.jsonl 2> $O/bench_extra.err; cat $O/bench_extra.jsonl
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_trace -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_trace.log 2>&1
tail -3 $O/prof_trace.log
timeout 900 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $O/prof_pmc_mfma -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_mfma.log 2>&1
tail -3 $O/prof_pmc_mfma.log
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --kernel-trace -d $O/prof_pmc_stall -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_stall.log 2>&1
timeout 900 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d $O/prof_pmc_fetch2 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_ifetch.log 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d $O/prof_pmc_cache -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_cache.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/prof_pmc_fetch -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/prof_pmc_write -o pmc -- python $R/bench.py --steps 2 --warmup 1 --batch 8192 --no-cpu-baseline > $O/prof_pmc_write.log 2>&1
cd $R
find $O -name '*.csv' | head -30
du -sh $O
