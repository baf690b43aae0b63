#!/bin/bash
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4m; rm -rf $O; mkdir -p $O; cd /tmp
export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip_ab.so
for v in 0 1; do
NELE_EIGH_INVIT_STORE=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$v -- python3 $GRAFT_REPO_ROOT/tools/eigh_time.py 256 420 3 > $O/log$v.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/s$v/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("%-70s calls %4s avg %9.1f us"%(r["Name"][:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
