#!/bin/bash
# Profile passes of the default bench workload (BASELINE configs[2]) on the current tree, on the GPU box:
#   kernel trace + stats (multi-stream and NELE_SERIAL=1), FETCH_SIZE and WRITE_SIZE in separate PMC passes, one SQ pass (MFMA busy).
# The bench runs with --no-isolated / no companions / no CPU leg: the kernel statistics hold the timed steps' own launches only
# (1 warm-up + 3 timed steps = 4 steps).  Summaries -> gpurun_out/<round>_prof/  (copy what is to be judged into profiles/<round>/)
ROUND=${1:-r04}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND}_prof; rm -rf $O; mkdir -p $O
ARGS="--steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
NELE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- python3 $R/bench.py $ARGS > $O/stats_serial.log 2>&1
if [ "${PMC:-1}" = "1" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/bench.py $ARGS > $O/pmc_$c.log 2>&1
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_SQ -- python3 $R/bench.py $ARGS > $O/pmc_SQ.log 2>&1
fi
tail -2 $O/stats.log $O/stats_serial.log
# the trace CSVs are large: keep the per-kernel summaries only (the PMC counter CSVs are needed by tools/make_traffic_json.py)
find $O -name "*kernel_trace.csv" -size +20M -delete
find $O -name "*_counter_collection.csv" -size +30M -exec python3 $R/tools/shrink_counters.py {} \;
ls -R $O | head -40
