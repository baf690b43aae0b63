#!/bin/bash
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/v2prof -- python3 $GRAFT_REPO_ROOT/scratch/siib_clean.py 32 64000 2>&1 | grep "SIIB clean"
cd "$GRAFT_REPO_ROOT"
python tools/kstats.py gpurun_out/v2prof 13 40 | cut -c1-150
python scratch/siib_clean.py 32 64000 | grep "SIIB clean"
python scratch/siib_clean.py 32 63871 | grep "SIIB clean"
python tools/eigh_time.py 32 420 5 2>&1 | tail -1
