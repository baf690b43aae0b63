#!/bin/bash
# r6 call r: a longer learning curve on the low-pass-noise corpus (100 epochs, both precisions) -> profiles/r06/learn_curve_lowpass_100.json
mkdir -p gpurun_out
timeout 1500 python tools/learn_curve.py --epochs 100 --noise-tilt 1.5 --out gpurun_out/learn_curve_lowpass_100.json > gpurun_out/r6r_learn.log 2>&1
tail -12 gpurun_out/r6r_learn.log
