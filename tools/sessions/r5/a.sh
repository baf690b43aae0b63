#!/bin/bash
# round 5, call a: baseline of this round's tree: GPU tests, the inference path alone (stages, plain, kernel stats)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for B in 64 128 256; do python tools/infer_time.py $B 20 stages; python tools/infer_time.py $B 30 plain; done
bash tools/prof_one.sh tools/infer_time.py 128 20 plain
cp gpurun_out/prof_one/*/*kernel_stats.csv gpurun_out/r5a_infer_kernel_stats.csv 2>/dev/null
