#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4j
python -m pytest tests/test_metrics_gpu.py -x -q -m gpu -k "eigensolver or repaired or siib" 2>&1 | tail -5 | tee gpurun_out/r4j/tests.txt
python tools/eigh_time.py 256 420 3 2>&1 | grep -v amdgpu | tee gpurun_out/r4j/eigh.txt
