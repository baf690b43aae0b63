#!/bin/bash
# Profile passes of the streamed inference path (BASELINE configs[4] per GPU: 8 s utterances, B = 128, 3 batches in flight):
# kernel stats, then the MFMA-side counters and FETCH_SIZE / WRITE_SIZE in separate PMC passes.  -> gpurun_out/<round>_infer/
ROUND=${1:-r05}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND}_infer; rm -rf $O; mkdir -p $O
cd /tmp
INFLIGHT=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/infer_time.py 128 40 stream > $O/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_plain -- python3 $R/tools/infer_time.py 128 20 plain > $O/stats_plain.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- python3 $R/tools/infer_time.py 128 10 plain > $O/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_SQ -- python3 $R/tools/infer_time.py 128 10 plain > $O/pmc_SQ.log 2>&1
grep -h "utt/s" $O/*.log
find $O -name "*kernel_trace.csv" -size +20M -delete
