#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4q
export TMPDIR=/tmp
for B in 256 32; do for v in "" .g0 .g1 .g3 .g2nl2 .g1nl2 .g2nl8; do
  export NELE_LIB=$GRAFT_REPO_ROOT/nele_gan_amd/libnele_hip.so$v
  cd /tmp; rm -rf /tmp/pq; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -- python3 $GRAFT_REPO_ROOT/tools/eigh_time.py $B 420 3 > /tmp/pq.log 2>&1
  cd $GRAFT_REPO_ROOT; echo "B=$B lib=$v $(python tools/kstats.py /tmp/pq 4 12 2>&1 | grep bisect) $(tail -1 /tmp/pq.log | sed 's/.*eigenvalue/eigenvalue/')"
done; done | tee gpurun_out/r4q/grid.txt
