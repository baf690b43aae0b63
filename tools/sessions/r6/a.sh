#!/bin/bash
# r6 call a: learning curves (f32, bf16) + the stand-alone eigensolver baseline of the round's starting tree
mkdir -p gpurun_out
python tools/eigh_time.py 256 420 5 > gpurun_out/r6a_eigh.log 2>&1
timeout 1500 python tools/learn_curve.py --epochs 30 --out gpurun_out/learn_curve_r6a.json > gpurun_out/r6a_learn.log 2>&1
tail -5 gpurun_out/r6a_eigh.log; tail -8 gpurun_out/r6a_learn.log
