#!/bin/bash
# one-device two-rank test + bench --gpus 2 on one device (gloo)
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -s 2>&1 | tail -15
NELE_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --global-batch 128 2> gpurun_out/r5q2_b2.err | tail -1 | cut -c1-1500
tail -5 gpurun_out/r5q2_b2.err
NELE_BENCH_ONE_DEVICE=1 timeout 1200 python bench.py --gpus 4 --steps 3 --warmup 1 2> gpurun_out/r5q2_b4.err | tail -1 | cut -c1-1500
tail -5 gpurun_out/r5q2_b4.err
