#!/bin/bash
# A/B of eigh.hip build variants (tools/variants.sh) on the eigensolver alone: kernel time of eigh_invit2_kernel + accuracy line
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4n; rm -rf $O; mkdir -p $O; cd /tmp
for v in "" .pf3 .pf3c8 ""; do
  export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$v -- python3 $GRAFT_REPO_ROOT/tools/eigh_time.py 256 420 3 > $O/log$v.txt 2>&1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("$O/s$v/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'invit' in r["Name"]: print("variant '$v' %-40s avg %9.1f us"%(r["Name"][:40],float(r["AverageNs"])/1e3))
PY
  grep "eigh B" $O/log$v.txt | cut -c1-200
done
