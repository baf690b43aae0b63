#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r4tl
cd /tmp; rm -rf /tmp/ptl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-utts 0 --companions 0 --no-isolated > /tmp/ptl.log 2>&1
f=$(ls /tmp/ptl/*/*kernel_trace.csv | head -1); [ -z "$f" ] && exit 1
cd $GRAFT_REPO_ROOT; python tools/timeline_summary.py "$f" 1.0 | tee gpurun_out/r4tl/timeline_b256.txt
