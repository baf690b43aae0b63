#!/bin/bash
# r6 call g: serialised kernel statistics of the step at the representative (non-periodic) length L = 63 871, and at the headline length
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; rm -rf $O; mkdir -p $O
ARGS="--steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated"
cd /tmp
NELE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/np_serial -- python3 $R/bench.py $ARGS --length 63871 > $O/np_serial.log 2>&1
NELE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/hl_serial -- python3 $R/bench.py $ARGS > $O/hl_serial.log 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head
tail -1 $O/np_serial.log | cut -c1-200
