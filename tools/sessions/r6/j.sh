#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6j; rm -rf $O; mkdir -p $O
cd /tmp
for c in 0 1; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c$c -- python3 $R/tools/tmp/epoch_kstats.py $c > $O/c$c.log 2>&1
grep "epochs" $O/c$c.log
done
find $O -name "*kernel_trace.csv" -delete
