#!/bin/bash
# Round-2 profile passes of the default bench workload (BASELINE configs[2]) on the current tree, on the GPU box:
#   kernel trace + stats, FETCH_SIZE and WRITE_SIZE in separate PMC passes, one SQ pass (MFMA busy).  Summaries -> gpurun_out/r02_prof/
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02_prof; rm -rf $O; mkdir -p $O
ARGS="--steps 3 --warmup 1 --cpu-utts 0 --companions 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
NELE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- python3 $R/bench.py $ARGS > $O/stats_serial.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py $ARGS > $O/pmc_$c.log 2>&1
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_SQ -- python3 $R/bench.py $ARGS > $O/pmc_SQ.log 2>&1
ls -R $O | head -40
# the trace CSVs are large: keep the per-kernel summaries only
find $O -name "*kernel_trace.csv" -size +20M -delete
