#!/bin/bash
# r6 call t: kernel times of the two-kernel IMCRA inside the enhancement path (128 x 8 s)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6t; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/infer_time.py 128 10 plain > $O/p.log 2>&1
find $O -name "*kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:16]:
    print('%-70s %5s %9.1f'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
