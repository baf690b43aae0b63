#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pw; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -- python3 $GRAFT_REPO_ROOT/tools/eigh_time.py ${1:-256} 420 3 > /tmp/pw.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/kstats.py /tmp/pw 4 12 2>&1 | grep eigh
