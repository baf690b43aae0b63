#!/bin/bash
cd /tmp; export TMPDIR=/tmp
NELE_SERIAL=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/t5prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-utts 0 --companions 0 --no-isolated > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/kstats.py gpurun_out/t5prof 4 120 | grep -E "gap|conv16_kernel<4|total" | cut -c1-150
