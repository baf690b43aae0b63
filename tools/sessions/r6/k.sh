#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6k; rm -rf $O; mkdir -p $O
for b in 8 32 64 256; do python3 tools/tmp/dstep.py $b 50 2>&1 | grep -v amdgpu; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d32 -- python3 $R/tools/tmp/dstep.py 32 50 > $O/d32.log 2>&1
NELE_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d32s -- python3 $R/tools/tmp/dstep.py 32 50 > $O/d32s.log 2>&1
find $O -name "*kernel_trace.csv" -delete
