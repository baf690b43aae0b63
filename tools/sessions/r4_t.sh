#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4t
python -m pytest tests/test_step_parity_gpu.py -x -q -m gpu -k "prefetch" 2>&1 | tail -5
python tools/pipe_time.py 32 2>&1 | grep ms/step | tee gpurun_out/r4t/pipe32.txt
python tools/pipe_time.py 64 'siib&haspi&estoi' 12 2>&1 | grep ms/step | tee gpurun_out/r4t/pipe64.txt
