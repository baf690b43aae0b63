#!/bin/bash
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4g; rm -rf $O; mkdir -p $O; cd /tmp
NELE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated > $O/log.txt 2>&1
cd $GRAFT_REPO_ROOT; find $O -name "*kernel_trace.csv" -size +20M -delete
python tools/kstats.py gpurun_out/r4g 4 70
