#!/bin/bash
# usage: prof_one.sh <script> [args]: rocprofv3 kernel stats of one scratch script, top 14 kernels
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_one; rm -rf $O; mkdir -p $O; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/"$@" > $O/log.txt 2>&1
grep -v "^[WE]2026" $O/log.txt | grep -v amdgpu.ids | tail -3
cd $R; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_one/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-64s calls %4s avg %10.1f us  %5s%%"%(r["Name"][:64],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
