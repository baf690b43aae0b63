#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4l
python -m pytest tests/test_metrics_gpu.py -x -q -m gpu -k "eigensolver or inverse_iteration" 2>&1 | tail -3 | tee gpurun_out/r4l/tests.txt
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4l; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $GRAFT_REPO_ROOT/tools/eigh_time.py 256 420 3 > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$O/s/**/*kernel_stats.csv",recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:6]:
    print("%-70s calls %4s avg %9.1f us"%(r["Name"][:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
grep "eigh B" $O/log.txt
