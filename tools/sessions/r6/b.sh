#!/bin/bash
# r6 call b: learning-curve test + the committed curve (two f32 seeds + bf16)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_learn_gpu.py -x -q -m gpu > gpurun_out/r6b_test.log 2>&1
timeout 1500 python tools/learn_curve.py --epochs 30 --f32-second-seed 667 --out gpurun_out/learn_curve_r6b.json > gpurun_out/r6b_learn.log 2>&1
tail -15 gpurun_out/r6b_test.log; grep -v " epoch " gpurun_out/r6b_learn.log | tail -8
