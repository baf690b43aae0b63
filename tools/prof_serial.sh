#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_serial; rm -rf $O; mkdir -p $O; cd /tmp
NELE_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-utts 0 --companions 0 > $O/log.txt 2>&1
grep -o '"ms_per_step": [0-9.]*' $O/log.txt
cd $R; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_serial/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f))); steps=4
print("sum per step %.2f ms"%(sum(float(r["TotalDurationNs"]) for r in rows)/1e6/steps))
for r in rows[:36]:
    print("%-66s n/step %5.1f avg %8.1f us  per-step %6.2f ms"%(r["Name"][:66],int(r["Calls"])/steps,float(r["AverageNs"])/1e3,float(r["TotalDurationNs"])/1e6/steps))
PY
